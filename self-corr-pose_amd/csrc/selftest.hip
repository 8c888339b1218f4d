// self-corr-pose_amd/csrc/selftest.hip -- device self-tests for the gfx950 packed-fp32 erratum this build is designed around.
//
// Measured on MI355X (profiles/r05_packed_fp32_erratum.txt, tools/pk_forms_probe2.py: all 51 op_sel / op_sel_hi / neg combinations):
//   v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 whose op_sel field is exactly [0,1] ([0,1,0] for fma: low result half = src0.lo (x) src1.HI)
//   return a WRONG LOW HALF in 0.6 % of the lanes while another wavefront of the same SIMD executes a K-doubled 16-bit matrix instruction
//   (v_mfma_f32_32x32x16_bf16 / _16x16x32_bf16 / _32x32x16_f16).  Every other op_sel combination, every op_sel_hi / neg combination, the
//   high half, the 16-bit packed instructions and the scalar VALU are never wrong; under v_mfma_f32_32x32x2_f32 nothing is wrong.
// hipcc's SLP vectoriser emits exactly that form for code like x1*y2 - x2*y1 (the rasteriser until round 4).  The build therefore switches
// the packed-fp32 feature off for every file that is not a bf16-MFMA GEMM (build.py NO_PACKED), tests/test_capi_symbols.py disassembles the
// library and fails on any op_sel:[0,1] packed-fp32 instruction, and these two entry points let tests/test_coresidency_gpu.py show on the
// box it runs on (a) that the screen's load does trigger the erratum (positive control) and (b) that the shipped kernels are clean under it.
// This file is the one place where the bad encoding appears on purpose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// register-only loop of ONE matrix instruction: no LDS, no memory traffic but the final store, ~40 VGPRs -- foreign wavefronts fit beside it
template <int KIND>
__global__ __launch_bounds__(256) void mfma_load_kernel(float* out, int iters, const int* stop) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    bf16x8 a8, b8;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a8[i] = (__bf16)((float)(lane + i) * 1e-3f);
        b8[i] = (__bf16)((float)(lane - i) * 1e-3f);
    }
    const float fa = lane * 1e-3f, fb = (63 - lane) * 1e-3f;
    for (int k = 0; k < iters; k++) {
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        if (KIND == 2) {          // the bf16 instruction in bursts: one MFMA, then the matrix pipe idles for ~450 cycles
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            __builtin_amdgcn_s_sleep(7);
        }
        // a persistent load: leave as soon as the host-side screen raises *stop (device memory or COHERENT host memory; polled every 16 384 instructions = ~0.25 ms -- 2 048 wavefronts polling host memory every 256 instructions saturated PCIe and starved every launch of the process, round 6; `iters` bounds the
        // launch whatever happens to the flag, so a failed test cannot leave the device spinning)
        if (stop != nullptr && (k & 16383) == 16383 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ float smul(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// every packed product is checked in place against the scalar instruction; counters[0] += wrong low halves, counters[1] += wrong high halves
template <int FORM>
__global__ __launch_bounds__(256) void packed_fp32_selftest_kernel(unsigned long long* counters, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float ax0 = 1.0f + (float)(i & 1023) * 9.765625e-4f, ay0 = 0.5f + (float)(i >> 10) * 1.220703125e-4f;
    const float bx0 = 1.25f + (float)(i & 511) * 1.953125e-3f, by0 = 0.75f + (float)(i >> 9) * 3.0517578125e-5f;
    unsigned lo = 0, hi = 0;
    for (int k = 0; k < iters; k++) {
        const float t = (float)(k & 255) * 3.90625e-3f;
        const f2 a = {ax0 + t, ay0 - t * 0.5f}, b = {bx0 - t * 0.25f, by0 + t};
        f2 r, e;
        if (FORM == 0) {          // the erratum form: low = a.lo * b.hi
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            e = (f2){smul(a.x, b.y), smul(a.y, b.x)};
        } else if (FORM == 1) {   // plain
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            e = (f2){smul(a.x, b.x), smul(a.y, b.y)};
        } else {                  // the mirrored selection (low = a.hi * b.lo): clean
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            e = (f2){smul(a.y, b.x), smul(a.x, b.y)};
        }
        lo += __float_as_uint(r.x) != __float_as_uint(e.x);
        hi += __float_as_uint(r.y) != __float_as_uint(e.y);
    }
    if (lo) atomicAdd(&counters[0], (unsigned long long)lo);
    if (hi) atomicAdd(&counters[1], (unsigned long long)hi);
}
}  // namespace

extern "C" int scp_selftest_mfma_load(int kind, float* out, int blocks, int iters, const int* stop, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kind == 0) hipLaunchKernelGGL(mfma_load_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters, stop);
    else if (kind == 1) hipLaunchKernelGGL(mfma_load_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters, stop);
    else if (kind == 2) hipLaunchKernelGGL(mfma_load_kernel<2>, dim3(blocks), dim3(256), 0, st, out, iters, stop);
    else return scp::fail(hipErrorInvalidValue, "scp_selftest_mfma_load: kind");
    return scp::check_launch("selftest_mfma_load");
}

extern "C" int scp_selftest_packed_fp32(int form, unsigned long long* counters, int blocks, int iters, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (form == 0) hipLaunchKernelGGL(packed_fp32_selftest_kernel<0>, dim3(blocks), dim3(256), 0, st, counters, iters);
    else if (form == 1) hipLaunchKernelGGL(packed_fp32_selftest_kernel<1>, dim3(blocks), dim3(256), 0, st, counters, iters);
    else if (form == 2) hipLaunchKernelGGL(packed_fp32_selftest_kernel<2>, dim3(blocks), dim3(256), 0, st, counters, iters);
    else return scp::fail(hipErrorInvalidValue, "scp_selftest_packed_fp32: form");
    return scp::check_launch("selftest_packed_fp32");
}
