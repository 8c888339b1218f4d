// self-corr-pose_amd/csrc/scp_common.h -- error plumbing shared by the kernels' host launchers.
#pragma once
#include <hip/hip_runtime.h>

namespace scp {
// records `what` (+ the HIP error text) for scp_last_error() and returns `code`
int fail(int code, const char* what);
// hipGetLastError() after a launch; 0 when clean
int check_launch(const char* what);

// scp::claim_vgprs<N>(): the kernel's wavefronts are allocated at least N vector registers (an empty asm statement that names
// v<N-1> as clobbered: no instruction, only the allocation size in the kernel descriptor).
// Why (round 4, tools/race_repro.py, profiles/r04_bf16_coresidency.txt): on this part a wavefront that shares a SIMD with a
// wavefront of a kernel issuing v_mfma_f32_32x32x16_bf16 can see its own vector registers change under it -- a trivial per-face kernel
// (csrc/softras.hip face_setup_kernel) returned wrong quotients for runs of ~10 wavefronts, the rasteriser wrong pixels, 30 of 60 passes,
// whenever the split-bf16 GEMMs of the frozen ViT ran on another stream; never next to the fp32-MFMA GEMMs, a library GEMM or with the
// streams serialised.  The kernels' register accounting is consistent (no AGPRs, no inline-asm registers), so this is handled as a
// hardware / firmware property: every kernel that issues bf16 MFMAs claims a register count that tiles the SIMD's 512-entry file
// exactly (2 x 256, 3 x 168, 4 x 128, 5 x 96), so that next to a full set of its wavefronts nothing else fits.  Its own occupancy is
// unchanged (the claimed size is the size class it already ran in).
template <int N> __device__ __forceinline__ void claim_vgprs();
#ifdef SCP_BF16_EXCLUSIVE     // probe build: 256 vector + 256 accumulation registers = the whole file, ONE wavefront per SIMD, nothing beside it ever
template <> __device__ __forceinline__ void claim_vgprs<96>() { asm volatile("" ::: "v255", "a255"); }
template <> __device__ __forceinline__ void claim_vgprs<128>() { asm volatile("" ::: "v255", "a255"); }
template <> __device__ __forceinline__ void claim_vgprs<168>() { asm volatile("" ::: "v255", "a255"); }
template <> __device__ __forceinline__ void claim_vgprs<256>() { asm volatile("" ::: "v255", "a255"); }
#else
template <> __device__ __forceinline__ void claim_vgprs<96>() { asm volatile("" ::: "v95"); }
template <> __device__ __forceinline__ void claim_vgprs<128>() { asm volatile("" ::: "v127"); }
template <> __device__ __forceinline__ void claim_vgprs<168>() { asm volatile("" ::: "v167"); }
template <> __device__ __forceinline__ void claim_vgprs<256>() { asm volatile("" ::: "v255"); }
#endif
// size class of a split-core GEMM tile with WM x WN MFMA tiles per wavefront (csrc/conv_igemm.hip, csrc/mutual_nn.hip)
template <int WM, int WN> __device__ __forceinline__ void claim_vgprs_for_tile() {
    if constexpr (WM * WN >= 8) claim_vgprs<256>();
    else if constexpr (WM * WN >= 4) claim_vgprs<168>();
    else if constexpr (WM * WN >= 2) claim_vgprs<128>();
    else claim_vgprs<96>();
}
}  // namespace scp
