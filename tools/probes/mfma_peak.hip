// tools/probes/mfma_peak.hip -- what the fp32 matrix pipe sustains on this device: pure MFMA loops (no memory traffic),
// v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, 1..4 wavefronts per SIMD, 2 or 4 independent accumulators.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return flops * 5 / (ms * 1e-3) / 1e12;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4 * 8 * 256 * 4);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps++) {          // wavefronts per SIMD: block = 4*wps wavefronts, one block per CU
        const int threads = 256 * wps, blocks = 256;
        double f32 = 2.0 * 32 * 32 * 2, f16 = 2.0 * 16 * 16 * 4;
        double waves = (double)blocks * threads / 64;
        printf("waves/SIMD %d: 32x32x2 acc4 %.1f  acc2 %.1f | 16x16x4 acc4 %.1f acc8 %.1f TFLOP/s\n", wps,
               run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, waves * iters * 8 * 4 * f32),
               run([&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, waves * iters * 8 * 2 * f32),
               run([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, waves * iters * 8 * 4 * f16),
               run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, waves * iters * 8 * 8 * f16));
    }
    return 0;
}
