"""self-corr-pose_amd/build.py -- compiles csrc/*.hip into the in-tree libscp_hip.so for gfx950.

    python self-corr-pose_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Flags that matter:
  -ffp-contract=off     parity: the reference semantics pinned by the oracle are un-contracted
  -munsafe-fp-atomics   native ds_add_f32 / global_atomic_add_f32 instead of CAS loops
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libscp_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-munsafe-fp-atomics",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function"]
# per-file extra flags
# -fno-slp-vectorize (rasteriser files): no packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 ...).  Round 4, DESIGN 5.2: with them the
# rasteriser kernels computed wrong values whenever another kernel on the device issued the K-doubled 16-bit MFMAs of gfx950
# (tools/race_repro.py: 60 of 60 passes next to a register-only bf16-MFMA loop; 0 of 60 built this way; not slower).
EXTRA = {"softras.hip": ["-ffp-contract=off", "-fno-slp-vectorize"] + (["-DSCP_FAST_GRAD_DIV"] if os.environ.get("SCP_FAST_GRAD_DIV") == "1" else []),
         "imgops.hip": ["-ffp-contract=off"], "softras_f64.hip": ["-ffp-contract=off", "-fno-slp-vectorize"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "scp_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(objdir, s[:-4] + ".o")
        src = os.path.join(CSRC, s)
        objs.append(o)
        headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]      # every .hip may include any of them
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(
                [os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "scp_hip.h")), os.path.getmtime(__file__)]
                + [os.path.getmtime(h) for h in headers]):
            continue
        cmd = [hipcc] + COMMON + EXTRA.get(s, []) + ["-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
