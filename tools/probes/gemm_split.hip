// tools/probes/gemm_split.hip -- NOT part of the library: stand-alone bench + accuracy check of the split-bf16 fp32 GEMM main loop
// (csrc/gemm_core_split.h) next to the fp32-MFMA main loop (csrc/gemm_core.h) on the ViT linear shapes, both against a float64
// reference computed on the GPU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I self-corr-pose_amd/csrc -I include tools/probes/gemm_split.hip -o tools/probes/gemm_split.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_core.h"
#include "gemm_core_split.h"

namespace {

template <class CFG, class CORE, class WT>
__global__ __launch_bounds__(CFG::THREADS, CFG::MINBLK) void gemm_kernel(const float* __restrict__ A, const WT* __restrict__ W,
                                                                         float* __restrict__ C, int M, int N, int K, int nblk_n,
                                                                         int per_xcd, int panels, long long* clk) {
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = blockIdx.x;
    const int lid = (t & 7) * per_xcd + (t >> 3);
    if (lid >= panels * nblk_n) return;
    const int bm = lid / nblk_n, bn = lid - bm * nblk_n;
    const int m0 = bm * CFG::BM, n0 = bn * CFG::BN;
    typename CORE::Acc acc;
    CORE core(lds);
    core.set_linear_sources(A, W, m0, n0, M, N, K);
    core.run(acc, K / CFG::BK);
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < CFG::WM; i++)
#pragma unroll
        for (int j = 0; j < CFG::WN; j++) {
            const int n = n0 + core.col_base() + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + core.row_base() + 32 * i + scp::acc_row(r, half);
                if (m < M && n < N) C[(size_t)m * N + n] = acc.t[i * CFG::WN + j][r];
            }
        }
    if (clk && (threadIdx.x & 63) == 0) {
        const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd((unsigned long long*)clk, (unsigned long long)(t1 - t0));
        atomicAdd((unsigned long long*)clk + 1, (unsigned long long)(r1 - r0));
    }
}

__global__ void naive_kernel(const float* A, const float* W, double* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double s = 0.0;
    for (int k = 0; k < K; k++) s += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    C[(size_t)m * N + n] = s;
}

__global__ void split_kernel(const float* W, __bf16* W3, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = W[i];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    W3[i] = h; W3[n + i] = m; W3[2 * n + i] = (__bf16)r2;
}

template <class CFG, class CORE, class WT>
void run(const char* name, const float* A, const WT* W, float* C, const std::vector<double>& ref, int M, int N, int K, std::vector<float>& h0) {
    const int lds_bytes = CFG::LDS_BYTES;
    auto kern = gemm_kernel<CFG, CORE, WT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, CFG::THREADS, lds_bytes);
    const int nblk_n = (N + CFG::BN - 1) / CFG::BN, panels = (M + CFG::BM - 1) / CFG::BM;
    const int per_xcd = (panels * nblk_n + 7) / 8, grid = per_xcd * 8;
    hipMemset(C, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, (long long*)nullptr);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); exit(1); }
    hipMemcpy(h0.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, scale = 0, sq = 0, sqr = 0;
    for (size_t i = 0; i < (size_t)M * N; i++) {
        const double d = (double)h0[i] - ref[i];
        maxerr = fmax(maxerr, fabs(d));
        scale = fmax(scale, fabs(ref[i]));
        sq += d * d; sqr += ref[i] * ref[i];
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, (long long*)nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, (long long*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    {
        long long* clk; hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, clk);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("    shader clock while this kernel runs: %.3f GHz\n", (double)h[0] / (double)h[1] * 0.1);
        hipFree(clk);
    }
    printf("%-30s M=%6d N=%5d K=%5d  %d WG/CU LDS %3d KB grid %5d: %7.1f us  %6.1f TFLOP/s(fp32-equiv)   max err %.2e rms err %.2e of rms %.2e (rel %.2e), scale %.2e\n",
           name, M, N, K, per_cu, lds_bytes / 1024, grid, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, maxerr, sqrt(sq / ((double)M * N)),
           sqrt(sqr / ((double)M * N)), sqrt(sq / sqr), scale);
    fflush(stdout);
}

}  // namespace

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32800;
    struct Shape { int N, K; const char* name; };
    const Shape shapes[] = {{1152, 384, "qkv"}, {384, 384, "proj"}, {1536, 384, "fc1"}, {384, 1536, "fc2"}};
    const size_t maxA = (size_t)M * 1536, maxW = (size_t)1536 * 1536, maxC = (size_t)M * 1536;
    float *A, *W, *C; double* Cref; __bf16* W3;
    hipMalloc(&A, maxA * 4); hipMalloc(&W, maxW * 4); hipMalloc(&C, maxC * 4); hipMalloc(&Cref, maxC * 8); hipMalloc(&W3, maxW * 6);
    std::vector<float> ha(maxA), hw(maxW), h0(maxC);
    std::vector<double> ref(maxC);
    srand(1);
    // activations with a wide dynamic range (like the residual stream), weights ~ N(0, 0.02)-ish
    for (auto& v : ha) { const float u = (float)rand() / RAND_MAX - 0.5f; v = u * expf(3.f * ((float)rand() / RAND_MAX - 0.5f)); }
    for (auto& v : hw) v = 0.08f * ((float)rand() / RAND_MAX - 0.5f);
    hipMemcpy(A, ha.data(), maxA * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), maxW * 4, hipMemcpyHostToDevice);
    for (const Shape& s : shapes) {
        const size_t nw = (size_t)s.N * s.K;
        hipLaunchKernelGGL(naive_kernel, dim3((s.N + 255) / 256, M), dim3(256), 0, 0, A, W, Cref, M, s.N, s.K);
        hipLaunchKernelGGL(split_kernel, dim3((nw + 255) / 256), dim3(256), 0, 0, W, W3, nw);
        hipDeviceSynchronize();
        hipMemcpy(ref.data(), Cref, (size_t)M * s.N * 8, hipMemcpyDeviceToHost);
        printf("--- %s\n", s.name);
        using F = scp::GemmCfg<4, 2, 2, 2, 2, 2>;
        run<F, scp::GemmCore<F>, float>("fp32 MFMA 256x128", A, W, C, ref, M, s.N, s.K, h0);
        using S = scp::SplitCfg<4, 2, 2, 2, 2>;
        run<S, scp::SplitGemmCore<S>, __bf16>("split bf16x6 256x128", A, W3, C, ref, M, s.N, s.K, h0);
        using S2 = scp::SplitCfg<2, 2, 2, 2, 2>;
        run<S2, scp::SplitGemmCore<S2>, __bf16>("split bf16x6 128x128", A, W3, C, ref, M, s.N, s.K, h0);
        using S3 = scp::SplitCfg<2, 4, 2, 2, 2>;
        run<S3, scp::SplitGemmCore<S3>, __bf16>("split bf16x6 128x256", A, W3, C, ref, M, s.N, s.K, h0);
        using S4 = scp::SplitCfg<2, 2, 4, 2, 2>;
        run<S4, scp::SplitGemmCore<S4>, __bf16>("split bf16x6 256x128 8 wavefronts", A, W3, C, ref, M, s.N, s.K, h0);
    }
    return 0;
}
