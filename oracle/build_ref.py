"""oracle/build_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

Builds oracle/_ref/ = the reference's own SoftRas kernels compiled here, from the sources where they lie under
/root/reference (nothing of them is copied into the repository):

    oracle/_ref/libref_softras.so             hipcc default flags (-ffp-contract=fast; fused multiply-adds, like the
                                              authors' nvcc build)
    oracle/_ref/libref_softras_nocontract.so  -ffp-contract=off (the un-contracted semantics the CPU oracle and the
                                              golden fixtures pin)

Recipe: lines 22-671 of third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu (the anonymous
namespace: device helpers + the three templated kernels on raw pointers) are written to a scratch file OUTSIDE the
repository and #included by oracle/ref_launcher.hip, which supplies hip_runtime.h and replays the reference's host
launchers (kernel.cu:674-813) on plain pointers.  The body compiles unchanged -- no stand-in headers, no edits.
Lines 1-21 (ATen / cuda.h includes, a pre-sm60 atomicAdd(double) fallback) and 674-813 (ATen launchers) are the parts
that need libtorch + CUDA headers and are not used.

On the GPU box /root/reference does not exist; the prebuilt .so files travel with the snapshot (oracle/_ref/ is
git-ignored, not gpurun-ignored) and this script is a no-op there.
"""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
REF_CU = "/root/reference/third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu"
FIRST, LAST = 22, 671          # `namespace{` ... matching `}` (checked below)
LAUNCHER = os.path.join(HERE, "ref_launcher.hip")
VARIANTS = {
    "libref_softras.so": [],
    "libref_softras_nocontract.so": ["-ffp-contract=off", "-DSCP_REF_CONTRACT_OFF"],
}


def available():
    return all(os.path.exists(os.path.join(OUT_DIR, n)) for n in VARIANTS)


def build(force=False, verbose=True):
    """returns the list of built libraries, or [] when the reference tree is absent (GPU box)"""
    if not os.path.exists(REF_CU):
        return [os.path.join(OUT_DIR, n) for n in VARIANTS] if available() else []
    os.makedirs(OUT_DIR, exist_ok=True)
    stale = [n for n in VARIANTS
             if force or not os.path.exists(os.path.join(OUT_DIR, n))
             or os.path.getmtime(os.path.join(OUT_DIR, n)) < max(os.path.getmtime(LAUNCHER), os.path.getmtime(REF_CU),
                                                                 os.path.getmtime(__file__))]
    if stale:
        with open(REF_CU) as fh:
            lines = fh.readlines()
        body = lines[FIRST - 1:LAST]
        assert body[0].strip() == "namespace{" and body[-1].strip() == "}", "reference kernel file changed"
        assert sum(l.count("__global__") for l in body) == 3
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        with tempfile.TemporaryDirectory(prefix="scp_ref_") as tmp:
            inc = os.path.join(tmp, "ref_kernel_body.inc")
            with open(inc, "w") as fh:
                fh.writelines(body)
            procs = []
            for name in stale:
                cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                       '-DSCP_REF_KERNEL_BODY="%s"' % inc] + VARIANTS[name] + [LAUNCHER, "-o", os.path.join(OUT_DIR, name)]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((name, subprocess.Popen(cmd)))
            for name, p in procs:
                if p.wait() != 0:
                    raise RuntimeError("hipcc failed building " + name)
    return [os.path.join(OUT_DIR, n) for n in VARIANTS]


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv)))
