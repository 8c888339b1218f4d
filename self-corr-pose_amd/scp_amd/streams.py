"""scp_amd/streams.py -- which kernels of a training step may share the device.

SCP_STREAMS=serial (default): one HIP stream.  Kernels of the step never run side by side.
SCP_STREAMS=overlap: the frozen-DINO ViT, the rotation-cycle branch (second encoder pass) and the soft-texture render pass run on side
streams, and Trainer.step() starts the NEXT batch's ViT pass during this step's backward (look-ahead).  ~6 ms per step faster at
B = 32 (29.9 vs 36.2 ms on one MI355X) -- and NOT safe on this part: a wavefront that shares a SIMD with a wavefront issuing
v_mfma_f32_32x32x16_bf16 (every split-bf16 GEMM / convolution / attention kernel of this build) can find its vector registers changed
under it.  Found in round 4 (tools/first_step_flake.py, tools/race_repro.py; DESIGN 5.2, profiles/r04_bf16_coresidency.txt): with the
ViT on a side stream the rasteriser returned different images for bit-identical inputs in 30 of 60 passes, a per-face elementwise kernel
wrong quotients for runs of ~600 faces, the step's loss terms moved by 1e-4 .. 5e-2 relative in one forward out of three; never with the
fp32 matrix cores, never with one stream, never when the bf16 kernels owned their SIMDs outright (a probe build, 2x slower).  Until the
cause is understood at the hardware / firmware level the shipped default keeps kernels of different streams apart."""
import os

MODE = os.environ.get("SCP_STREAMS", "serial")
if MODE not in ("serial", "overlap"):
    raise ValueError("SCP_STREAMS must be 'serial' or 'overlap', not %r" % MODE)


def overlap():
    return MODE == "overlap"
