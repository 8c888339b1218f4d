// tools/probes/gemm_v3.hip -- NOT part of the library: stand-alone bench + check of the software-pipelined fp32 GEMM main loop
// (csrc/gemm_core.h) on the four ViT linear shapes, against a naive GPU reference.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I self-corr-pose_amd/csrc -I include tools/probes/gemm_v3.hip -o tools/probes/gemm_v3.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_core.h"

namespace {

template <class CFG>
__global__ __launch_bounds__(CFG::THREADS, CFG::MINBLK) void gemm_plain_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                                               float* __restrict__ C, int M, int N, int K, int nblk_n,
                                                                               int per_xcd, int panels, long long* stamps) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using Core = scp::GemmCore<CFG>;
    const int t = blockIdx.x;
    const int lid = (t & 7) * per_xcd + (t >> 3);
    if (lid >= panels * nblk_n) return;
    const int bm = lid / nblk_n, bn = lid - bm * nblk_n;
    const int m0 = bm * CFG::BM, n0 = bn * CFG::BN;
    typename Core::Acc acc;
    Core core(lds);
    core.set_linear_sources(A, W, m0, n0, M, N, K);
    const long long t1 = __builtin_amdgcn_s_memtime();
    core.run(acc, K / CFG::BK);
    const long long t2 = __builtin_amdgcn_s_memtime();
    // epilogue: plain store
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < CFG::WM; i++)
#pragma unroll
        for (int j = 0; j < CFG::WN; j++) {
            const int n = n0 + core.col_base() + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + core.row_base() + 32 * i + scp::acc_row(r, half);
                if (m < M && n < N && (!stamps || acc.t[i * CFG::WN + j][r] == 12345.678f)) C[(size_t)m * N + n] = acc.t[i * CFG::WN + j][r];
            }
        }
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t3 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) {
            long long* o = stamps + ((size_t)blockIdx.x * CFG::NW + (threadIdx.x >> 6)) * 4;
            long long* clk = stamps + (size_t)gridDim.x * CFG::NW * 4 + ((size_t)blockIdx.x * CFG::NW + (threadIdx.x >> 6)) * 2;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
            const long long r3 = __builtin_amdgcn_s_memrealtime();
            clk[0] = t3 - t0; clk[1] = r3 - r0;
        }
    }
}

__global__ void naive_kernel(const float* A, const float* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    double s = 0.0;
    for (int k = 0; k < K; k++) s += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    C[(size_t)m * N + n] = (float)s;
}

template <class CFG>
void run(const char* name, const float* A, const float* W, float* C, const float* Cref, int M, int N, int K, std::vector<float>& h0,
         std::vector<float>& h1) {
    const int lds_bytes = CFG::LDS_BYTES;
    auto kern = gemm_plain_kernel<CFG>;
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
    hipMemcpy(h1.data(), Cref, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, scale = 0;
    for (size_t i = 0; i < (size_t)M * N; i++) {
        maxerr = fmax(maxerr, fabs((double)h0[i] - (double)h1[i]));
        scale = fmax(scale, fabs((double)h1[i]));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, (long long*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    {
        long long* st;
        const size_t ns = (size_t)grid * CFG::NW * 4;
        hipMalloc(&st, ns * 8 * 2); hipMemset(st, 0, ns * 8 * 2);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, st);
        hipDeviceSynchronize();
        std::vector<long long> hs(ns * 2);
        hipMemcpy(hs.data(), st, ns * 8 * 2, hipMemcpyDeviceToHost);
        double ct = 0, cr = 0;
        for (size_t w = 0; w < ns / 4; w++) { ct += hs[ns + 2 * w]; cr += hs[ns + 2 * w + 1]; }
        printf("    shader clock during the launch: %.3f GHz (s_memtime / s_memrealtime at 100 MHz)\n", ct / cr * 0.1);
        hipFree(st);
        double pro = 0, loop = 0, epi = 0; long long tmin = -1, tmax = 0; int cnt = 0;
        for (size_t w = 0; w < ns / 4; w++) {
            if (hs[4 * w + 3] == 0) continue;
            pro += hs[4 * w + 1] - hs[4 * w]; loop += hs[4 * w + 2] - hs[4 * w + 1]; epi += hs[4 * w + 3] - hs[4 * w + 2];
            if ((w / CFG::NW) % 8 == 0) {                       // one XCD's workgroups share a counter
                if (tmin < 0 || hs[4 * w] < tmin) tmin = hs[4 * w];
                if (hs[4 * w + 3] > tmax) tmax = hs[4 * w + 3];
            }
            cnt++;
        }
        hipEvent_t g0, g1; hipEventCreate(&g0); hipEventCreate(&g1);
        hipEventRecord(g0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, (long long*)nullptr);
        hipEventRecord(g1); hipEventSynchronize(g1);
        float gms; hipEventElapsedTime(&gms, g0, g1);
        printf("    per wavefront (memtime ticks): setup %.0f  main loop %.0f (ideal %d)  epilogue %.0f ; XCD0 span %lld ticks for a %.1f us launch = %.2f ticks/ns\n", pro / cnt, loop / cnt,
               (K / 16) * CFG::NM * 2 * 64, epi / cnt, tmax - tmin, gms * 1e3, (tmax - tmin) / (gms * 1e6));
    }
    {
        long long* st;
        const size_t ns = (size_t)grid * CFG::NW * 4;
        hipMalloc(&st, ns * 8 * 2);
        hipEvent_t f0, f1; hipEventCreate(&f0); hipEventCreate(&f1);
        hipEventRecord(f0);
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, st);
        hipEventRecord(f1); hipEventSynchronize(f1);
        float ms2; hipEventElapsedTime(&ms2, f0, f1);
        printf("    without the C stores: %7.1f us  %6.1f TFLOP/s\n", ms2 / reps * 1e3, 2.0 * M * N * K / (ms2 / reps * 1e-3) / 1e12);
        hipFree(st);
    }
    printf("%-28s M=%6d N=%5d K=%5d  %d WG/CU LDS %3d KB grid %5d: %7.1f us  %6.1f TFLOP/s   max err %.2e of scale %.2e\n", name, M, N, K, per_cu,
           lds_bytes / 1024, grid, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, maxerr, scale);
    fflush(stdout);
}

}  // namespace

template <class CFG>
void race(const char* name, const float* A, const float* W, float* C, const float* Cref, int M, int N, int K, std::vector<float>& h0, std::vector<float>& h1) {
    const int lds_bytes = CFG::LDS_BYTES;
    auto kern = gemm_plain_kernel<CFG>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int nblk_n = (N + CFG::BN - 1) / CFG::BN, panels = (M + CFG::BM - 1) / CFG::BM;
    const int per_xcd = (panels * nblk_n + 7) / 8, grid = per_xcd * 8;
    hipMemcpy(h1.data(), Cref, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int rep = 0; rep < 40; rep++) {
        hipMemset(C, 0, (size_t)M * N * 4);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(CFG::THREADS), lds_bytes, 0, A, W, C, M, N, K, nblk_n, per_xcd, panels, (long long*)nullptr);
        hipMemcpy(h0.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        double e = 0;
        for (size_t i = 0; i < (size_t)M * N; i++) e = fmax(e, fabs((double)h0[i] - (double)h1[i]));
        if (e > 1e-3) bad++;
        worst = fmax(worst, e);
    }
    printf("race %-24s M=%d N=%d K=%d: %d bad launches of 40, worst err %.3e\n", name, M, N, K, bad, worst);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32800;
    const bool race_mode = argc > 2;
    struct Shape { int N, K; const char* name; };
    const Shape shapes[] = {{1152, 384, "qkv"}, {384, 384, "proj"}, {1536, 384, "fc1"}, {384, 1536, "fc2"}, {128, 1152, "conv128"}, {64, 576, "conv64"}, {1024, 384, "n1024"}};
    const size_t maxA = (size_t)M * 1536, maxW = (size_t)1536 * 1536, maxC = (size_t)M * 1536;
    float *A, *W, *C, *Cref;
    hipMalloc(&A, maxA * 4); hipMalloc(&W, maxW * 4); hipMalloc(&C, maxC * 4); hipMalloc(&Cref, maxC * 4);
    std::vector<float> ha(maxA), hw(maxW), h0(maxC), h1(maxC);
    srand(1);
    for (auto& v : ha) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hw) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, ha.data(), maxA * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), maxW * 4, hipMemcpyHostToDevice);
    if (race_mode) {
        for (int K : {32, 64, 96, 384}) for (int N : {128, 384}) {
            hipLaunchKernelGGL(naive_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, W, Cref, M, N, K);
            hipDeviceSynchronize();
            if (K % 32 == 0) race<scp::GemmCfg<4, 2, 2, 2, 2, 2>>("256x128 s2", A, W, C, Cref, M, N, K, h0, h1);
            if (K % 32 == 0) race<scp::GemmCfg<1, 2, 2, 2, 2, 2>>("64x128 s2", A, W, C, Cref, M, N, K, h0, h1);
            if (K % 48 == 0) race<scp::GemmCfg<4, 2, 2, 2, 3, 2>>("256x128 s3", A, W, C, Cref, M, N, K, h0, h1);
            if (K % 48 == 0) race<scp::GemmCfg<1, 2, 2, 2, 3, 2>>("64x128 s3", A, W, C, Cref, M, N, K, h0, h1);
        }
        return 0;
    }
    for (const Shape& s : shapes) {
        hipLaunchKernelGGL(naive_kernel, dim3((s.N + 255) / 256, M), dim3(256), 0, 0, A, W, Cref, M, s.N, s.K);
        hipDeviceSynchronize();
        printf("--- %s\n", s.name);
        //                       WM WN NWM NWN NSTAGE MINBLK
        if (s.N % 128 == 0) {
            run<scp::GemmCfg<4, 2, 2, 2, 3, 2>>("256x128 w128x64 s3 2/CU", A, W, C, Cref, M, s.N, s.K, h0, h1);
            run<scp::GemmCfg<4, 2, 2, 2, 2, 2>>("256x128 w128x64 s2 2/CU", A, W, C, Cref, M, s.N, s.K, h0, h1);
            run<scp::GemmCfg<2, 2, 2, 2, 3, 3>>("128x128 w64x64 s3 3/CU", A, W, C, Cref, M, s.N, s.K, h0, h1);
            run<scp::GemmCfg<2, 2, 2, 2, 2, 3>>("128x128 w64x64 s2 3/CU", A, W, C, Cref, M, s.N, s.K, h0, h1);
            run<scp::GemmCfg<2, 4, 2, 1, 3, 2>>("128x128 w64x128 2wv s3", A, W, C, Cref, M, s.N, s.K, h0, h1);
        }
        run<scp::GemmCfg<4, 2, 2, 1, 3, 2>>("256x64 w128x64 2wv s3", A, W, C, Cref, M, s.N, s.K, h0, h1);
        run<scp::GemmCfg<2, 2, 4, 1, 3, 3>>("256x64 w64x64 4wv s3", A, W, C, Cref, M, s.N, s.K, h0, h1);
        run<scp::GemmCfg<4, 1, 2, 2, 3, 3>>("256x64 w128x32 4wv s3", A, W, C, Cref, M, s.N, s.K, h0, h1);
    }
    return 0;
}
