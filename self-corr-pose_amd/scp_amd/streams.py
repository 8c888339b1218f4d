"""scp_amd/streams.py -- which kernels of a training step may share the device.

SCP_STREAMS=overlap (default): the frozen-DINO ViT, the rotation-cycle branch (second encoder pass) and the soft-texture render pass run
on side HIP streams, Trainer.step() starts the NEXT batch's ViT pass during this step's backward (look-ahead) and the gradient buckets
are all-reduced from inside backward.  ~6 ms per step faster at B = 32 than one stream (30.0 vs 36.2 ms on one MI355X).
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
  * tests/test_coresidency_gpu.py screens every stage of the B = 32 step (forward AND backward, clip + fused AdamW, a 1-rank RCCL
    all-reduce) and the whole step under a persistent bf16-MFMA load, bit-exact where the stage is deterministic, with two positive
    controls that must FAIL on the box the test runs on (the self-checking erratum kernel; the rasteriser as compiled until round 4).
"""
import os

MODE = os.environ.get("SCP_STREAMS", "overlap")
if MODE not in ("serial", "overlap"):
    raise ValueError("SCP_STREAMS must be 'serial' or 'overlap', not %r" % MODE)


def overlap():
    return MODE == "overlap"
