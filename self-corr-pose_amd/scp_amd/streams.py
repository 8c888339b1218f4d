"""scp_amd/streams.py -- which kernels of a training step may share the device.

SCP_STREAMS=overlap (default): the frozen-DINO ViT and the rotation-cycle branch (second encoder pass) run on side HIP streams of the
step's own (side_stream below), Trainer.step() starts the NEXT batch's ViT pass during this step's backward (look-ahead) and the gradient
buckets are all-reduced from inside backward.  ~6 ms per step faster at B = 32 than one stream (29.2-30.5 vs 36.2 ms on one MI355X).
The soft-texture render pass on a THIRD side stream (-0.4 ms) is opt-in, SCP_TEXTURE_STREAM=1: see overlap_texture() for why.
SCP_STREAMS=serial: one HIP stream; kernels of the step never run side by side.

Why this was `serial` for half a round, and why it no longer has to be.  Round 4 found kernels returning different results
for bit-identical inputs whenever they shared the device with the split-bf16 GEMM / convolution / attention kernels.  Round 5 pinned it
down to ONE instruction form of gfx950: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 whose op_sel field is [0,1] (low half = src0.lo (x)
src1.hi) return a wrong low half in ~0.6 % of the lanes while a K-doubled 16-bit MFMA (v_mfma_f32_32x32x16_bf16 & co) executes on the same
SIMD -- every other op_sel / op_sel_hi / neg combination, the high half, 16-bit packed ops and scalar VALU are exact (self-checking
probes over all 51 combinations, profiles/r05_packed_fp32_erratum.txt).  hipcc's SLP vectoriser emits that form for cross products
(x1*y2 - x2*y1: the rasteriser's per-face kernel).  What keeps the overlapped schedule safe now:
  * every file of the library that is not itself a bf16 GEMM is compiled with the packed-fp32 feature switched off (build.py NO_PACKED) and
    a CPU test disassembles the shipped code objects: no packed-fp32 instruction with ANY op_sel in 220 kernels
    (tests/test_capi_symbols.py);
  * of the 110 ATen / rocprim kernels a step launches none carries the form (libtorch_hip.so scanned: 767 of its kernels do, none of them on
    this path; profiles/r05_torch_kernel_scan.txt) -- re-scan when torch is upgraded (tools/packed_census.py);
  * the fp32 GEMM kernels rocBLAS / hipBLASLt can serve the step's remaining library products with (20 169 kernels in 30 gfx950 code
    objects, helper kernels included) carry the form nowhere but in rocBLAS's complex-single PostGSU reduction, which no fp32 GEMM
    launches (tools/library_gemm_scan.py, profiles/r06_library_gemm_scan.txt); the scan is stamped with the torch / HIP version and a
    digest of those files, and an installation the stamp does not describe DEFAULTS TO `serial` (stamp_status below) -- round 6, ADVICE;
  * tests/test_coresidency_gpu.py screens every stage of the B = 32 step (forward AND backward, clip + fused AdamW, a 1-rank RCCL
    all-reduce) and the whole step under a persistent bf16-MFMA load, bit-exact where the stage is deterministic, with two positive
    controls that must FAIL on the box the test runs on (the self-checking erratum kernel; the rasteriser as compiled until round 4).
"""
import os

STAMP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "coresidency_stamp.json")


def stamp_status():
    """(ok, why): does the static scan this build ships (tuning/coresidency_stamp.json: torch / HIP version and a digest over the names
    and sizes of the gfx950 fp32 rocBLAS / hipBLASLt code objects, written by tools/library_gemm_scan.py) describe the installation
    that is running?  Which library GEMM kernel serves a call is the library's choice (heuristics, TunableOp), so the rule "no kernel
    of the step carries the erratum form" can only be vouched for per library build: a different torch / ROCm needs a re-scan
    (ADVICE r5).  No GPU and no torch.cuda call needed."""
    import hashlib
    import json
    try:
        with open(STAMP) as f:
            want = json.load(f)
    except (OSError, ValueError) as e:
        return False, "no readable stamp (%s)" % (e,)
    import torch
    if (torch.__version__, str(torch.version.hip)) != (want.get("torch"), want.get("hip")):
        return False, "scanned torch %s / HIP %s, running torch %s / HIP %s" % (want.get("torch"), want.get("hip"), torch.__version__,
                                                                                torch.version.hip)
    lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    h, n = hashlib.sha256(), 0
    for sub in ("rocblas/library", "hipblaslt/library"):
        d = os.path.join(lib, sub)
        for f in sorted(os.listdir(d)) if os.path.isdir(d) else ():
            if "gfx950" in f and (f.endswith(".co") or f.endswith(".hsaco")) and (f.startswith("Kernels.so") or "Type_SS_" in f):
                h.update(("%s:%d;" % (f, os.path.getsize(os.path.join(d, f)))).encode())
                n += 1
    if (n, h.hexdigest()[:16]) != (want.get("code_objects"), want.get("digest")):
        return False, "the library GEMM code objects differ from the scanned ones (%d files, digest %s)" % (n, h.hexdigest()[:16])
    return True, "scan matches torch %s / HIP %s" % (want["torch"], want["hip"])


def _default_mode():
    """`overlap` only for an installation the shipped scan vouches for; anything else gets the one-stream schedule and one line that
    says why (SCP_STREAMS=overlap overrides: the dynamic screen, tests/test_coresidency_gpu.py, is then the only evidence)"""
    ok, why = stamp_status()
    if ok:
        return "overlap"
    import warnings
    warnings.warn("scp_amd.streams: SCP_STREAMS defaults to 'serial' here -- %s; re-run tools/library_gemm_scan.py and "
                  "tools/packed_census.py for this installation (DESIGN 5.2), or set SCP_STREAMS=overlap" % why)
    return "serial"


MODE = os.environ.get("SCP_STREAMS") or _default_mode()
if MODE not in ("serial", "overlap"):
    raise ValueError("SCP_STREAMS must be 'serial' or 'overlap', not %r" % MODE)


def overlap():
    return MODE == "overlap"


def overlap_texture():
    """the soft-texture render pass + its loss on a THIRD side stream?  Off by default since round 6 (SCP_TEXTURE_STREAM=1 switches it on).
    With three side streams (frozen ViT, rotation-cycle pass, texture pass) the step loop has stream placements at which the device stops
    -- every stream waiting on an event, no wavefront resident; with ROC_CPU_WAIT_FOR_SIGNAL=1 the host blocks instead -- in 60-100 % of
    fresh processes: torch pool entries (3,4,5) / (4,5,6) after other users of the pool, the step's own streams as the 5th-7th or 8th-10th
    stream of the process, with 4 or 16 hardware queues alike (profiles/r06_stall_rates_call4..8.txt, ~600 processes of
    tools/r06/hang_repro.py; this is what hung round 5's GPU suite).  With the texture pass on the main stream -- two side streams -- 0 of
    98 processes stalled over 14 placements, pool and own streams (profiles/r06_stall_rates_call10.txt); the cause sits below this code
    (torch 2.10.0+rocm7.0 / HIP 7.0.51831) and is not understood, so the default is the schedule that has no known stalling placement.
    Cost: DESIGN 5.4."""
    return overlap() and os.environ.get("SCP_TEXTURE_STREAM", "0") == "1"


class DeviceStall(RuntimeError):
    """the device did not finish work the host is waiting for within the bound (see wait_bounded)"""


def stall_timeout():
    """seconds a host-side wait of the training / test loops may take before it raises (SCP_DEVICE_TIMEOUT_S, default 300; 0 = wait forever)"""
    return float(os.environ.get("SCP_DEVICE_TIMEOUT_S", "300"))


def wait_bounded(event, what, named_streams=None, timeout=None):
    """host wait for `event` that cannot stall a process for ever: polls the event and, past the bound, raises DeviceStall naming the
    streams that are still busy.  The loops' only host<->device synchronisation points (loss read-back at log time) go through this, so
    a device that stops completing work (round 5's GPU suite hung in exactly such a read-back, with no record of what was pending)
    becomes an error that says where, instead of a wait inside hipStreamSynchronize that no signal handler can interrupt."""
    import time
    timeout = stall_timeout() if timeout is None else timeout
    t0, pause = time.monotonic(), 5e-5
    while not event.query():
        if timeout > 0 and time.monotonic() - t0 > timeout:
            busy = []
            for name, s in (named_streams() if callable(named_streams) else (named_streams or {})).items():
                try:
                    if s is not None and not s.query():
                        busy.append(name)
                except Exception as e:                      # noqa: BLE001 -- the report must not raise something else
                    busy.append("%s (query failed: %r)" % (name, e))
            where = crumb_report(named_streams)
            raise DeviceStall("%s: the device did not finish within %.0f s; busy streams: %s%s" % (what, timeout, ", ".join(busy) or "none",
                                                                                                 ("; crumbs: " + where) if where else ""))
        time.sleep(pause)
        pause = min(pause * 1.5, 2e-3)


# ---- optional breadcrumbs: where on each stream did the device stop? -----------------------------------------------------------------
# SCP_CRUMBS=1: Trainer.step / MeshNet.forward / the ViT prefetch record a named event on their stream at every phase boundary; a
# DeviceStall then lists, per stream, the last crumb the device reached and the first it did not.  Off by default: an event record is a
# marker packet in the stream's queue, and the uninstrumented step is the product.
CRUMBS = os.environ.get("SCP_CRUMBS", "0") == "1"
_crumbs = []


def crumb(name, stream=None):
    if not CRUMBS:
        return
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(s)
    _crumbs.append((name, s.cuda_stream, ev))
    if len(_crumbs) > 6000:
        del _crumbs[:3000]


def crumb_report(named_streams=None):
    """per stream: 'last reached -> first not reached' over the recorded crumbs (empty string when crumbs are off)"""
    if not _crumbs:
        return ""
    names = {}
    for k, s in ((named_streams() if callable(named_streams) else (named_streams or {})).items()):
        if s is not None:
            names[s.cuda_stream] = k
    per = {}
    for name, sid, ev in _crumbs:
        per.setdefault(sid, []).append((name, ev))
    out = []
    for sid, lst in per.items():
        done = [ev.query() for _, ev in lst]
        first_pending = next((i for i, d in enumerate(done) if not d), None)
        if first_pending is None:
            out.append("%s: all %d crumbs reached (last: %s)" % (names.get(sid, hex(sid)), len(lst), lst[-1][0]))
        else:
            out.append("%s: reached %s, NOT reached %s (%d pending)" % (names.get(sid, hex(sid)), lst[first_pending - 1][0] if first_pending else "<nothing>",
                                                                        lst[first_pending][0], len(lst) - first_pending))
    return "; ".join(out)


# ---- the step's own side streams ------------------------------------------------------------------------------------------------------
# SCP_SIDE_STREAMS=own (default): a side stream is a HIP stream this build creates for the purpose (scp_stream_create), wrapped as a
# torch.cuda.ExternalStream -- a fresh stream nobody used before.  SCP_SIDE_STREAMS=pool: torch.cuda.Stream(), i.e. the next entry of
# torch's round-robin pool of 32 long-lived streams per device, whose earlier users (other libraries, earlier phases of the process) and
# hardware-queue placement the step knows nothing about.  Round 6: with the three side streams of the step on pool entries
# that had been used before, the step loop stalled on the device -- every stream waiting, no wavefront resident -- in 24 of 30 fresh
# processes at one pool position and in none at any other; the GPU suite hit that position whenever test_coresidency_gpu.py had run first.
SIDE_STREAMS = os.environ.get("SCP_SIDE_STREAMS", "own")
_own_streams = []          # (handle, wrapper): kept for the life of the process -- a stream the autograd graph may still name is never destroyed


def _new_own_stream(index):
    import ctypes
    import torch
    from . import capi
    handle = ctypes.c_void_p()
    with torch.cuda.device(index):
        capi.check(capi.lib().scp_stream_create(ctypes.byref(handle)), "scp_stream_create")
    s = torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", index))
    _own_streams.append((handle.value, s))
    return s


def side_stream(device):
    """a stream for one overlapped branch of the step (frozen ViT, rotation-cycle pass, texture pass, gradient all-reduce)"""
    import torch
    if SIDE_STREAMS != "own":
        return torch.cuda.Stream(device=device)
    dev = torch.device(device)
    return _new_own_stream(dev.index if dev.index is not None else torch.cuda.current_device())
