// tools/probes/mfma_aggressor.hip -- which matrix instruction disturbs its SIMD neighbours?  (tools/race_repro.py)
// Register-only kernels: every wavefront issues `iters` MFMAs of ONE kind on constant operands (no LDS, no memory traffic but one
// store at the end), ~40 vector registers, so that foreign wavefronts fit beside them on every SIMD.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/mfma_aggressor.hip -o tools/probes/libmfma_aggr.so
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8;
    f16x8 ha, hb;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a8[i] = (__bf16)(seed * (float)(lane + i) * 1e-3f);
        b8[i] = (__bf16)(seed * (float)(lane - i) * 1e-3f);
        ha[i] = (_Float16)(seed * (float)(lane + i) * 1e-3f);
        hb[i] = (_Float16)(seed * (float)(lane - i) * 1e-3f);
    }
    const float fa = seed * lane * 1e-3f, fb = seed * (63 - lane) * 1e-3f;
    for (int k = 0; k < iters; k++) {
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        if (KIND == 1) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc4, 0, 0, 0);
        if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        if (KIND == 3) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        if (KIND == 5) {          // the bf16 instruction at a fraction of its rate: one MFMA (32 cycles of the pipe), then ~64 * 7 idle cycles
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            __builtin_amdgcn_s_sleep(7);
        }
        if (KIND == 6) {          // ... at about half rate
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
            asm volatile("s_nop 15");
            asm volatile("s_nop 15");
        }
        if (KIND == 4) {          // no matrix instruction: the same loop on the vector ALU
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = __builtin_fmaf(acc[r], 0.999f, fa);
        }
    }
    float s = acc4[0] + acc4[1] + acc4[2] + acc4[3];
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int mfma_aggressor(int kind, float* out, int blocks, int iters, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (kind) {
        case 0: hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
        case 1: hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
        case 2: hipLaunchKernelGGL(aggressor<2>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
        case 3: hipLaunchKernelGGL(aggressor<3>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
        case 5: hipLaunchKernelGGL(aggressor<5>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
        case 6: hipLaunchKernelGGL(aggressor<6>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
        default: hipLaunchKernelGGL(aggressor<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f); break;
    }
    return (int)hipGetLastError();
}

// ---- victims: deterministic register-only loops, one ingredient each; out[i] depends on i only --------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void victim(float* out, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float x = 1.0f + (float)(i & 1023) * 1e-3f, y = 0.5f + (float)(i >> 10) * 1e-4f;
    for (int k = 0; k < iters; k++) {
        if (KIND == 0) x = x / (y + 1.0f) + 1.25f;                                   // IEEE fp32 division (v_div_scale / v_div_fmas / v_div_fixup)
        if (KIND == 1) x = (float)(1.0 / (1.0 + (double)x * 0.5)) + y;               // conversions + double division
        if (KIND == 2) { if (((i + k) & 3) == 0) x = x * 0.99f + y; else if (x > 1.5f) x = x * 0.5f + 0.75f; else x = x * 1.01f + 0.01f; }   // divergence
        if (KIND == 3) x = __expf(-x * 0.1f) + sqrtf(x + y) * 0.5f + __frcp_rn(x + 2.0f);   // transcendental unit
        if (KIND == 4) x = __builtin_fmaf(x, 0.999f, y * 0.001f);                     // plain fma chain
        if (KIND == 6) {          // packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: VOP3P, the encoding family of the MFMAs)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 v = {x, y};
            const f32x2 c = {0.999f, 1.001f}, d = {0.001f, -0.0005f};
            v = v * c + d;
            v = v * v.yx * (f32x2){0.5f, 0.5f} + (f32x2){0.25f, 0.25f};
            x = v.x; y = v.y;
        }
        if (KIND == 5) { const float m = fmaxf(x, y), n = fminf(x, y); x = (m > 1.2f ? m * 0.9f : m + 0.1f) + (n < 0.6f ? 0.05f : -0.01f); }   // compares / selects
    }
    out[i] = x;
}

extern "C" int probe_victim(int kind, float* out, int blocks, int iters, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (kind) {
        case 0: hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), 0, st, out, iters); break;
        case 1: hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), 0, st, out, iters); break;
        case 2: hipLaunchKernelGGL(victim<2>, dim3(blocks), dim3(256), 0, st, out, iters); break;
        case 3: hipLaunchKernelGGL(victim<3>, dim3(blocks), dim3(256), 0, st, out, iters); break;
        case 4: hipLaunchKernelGGL(victim<4>, dim3(blocks), dim3(256), 0, st, out, iters); break;
        case 6: hipLaunchKernelGGL(victim<6>, dim3(blocks), dim3(256), 0, st, out, iters); break;
        default: hipLaunchKernelGGL(victim<5>, dim3(blocks), dim3(256), 0, st, out, iters); break;
    }
    return (int)hipGetLastError();
}
