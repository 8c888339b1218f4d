// tools/probes/gemm_shapes.hip -- which workgroup / wavefront tiling feeds the fp32 matrix pipe best through LDS-DMA?  The
// main loop of csrc/vit_gemm.hip without addresses that mean anything: per chunk every wavefront issues its share of the
// (BM + BN) x BK operand fill as 1-KiB LDS-DMA instructions (16 rows x 64 B gather, row pitch 1536 B), waits for the chunk
// two behind (STAGES-1 in flight), s_barrier, reads its fragments with ds_read_b128 and runs WM x WN x (BK/2) MFMA 32x32x2.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gemm_shapes.hip -o tools/probes/gemm_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

template <int WM, int WN, int NWM, int NWN, int BK, int STAGES, int MINBLK>
__global__ __launch_bounds__(64 * NWM * NWN, MINBLK) void skel(const float* src, unsigned src_mask, float* out, int chunks) {
    constexpr int NW = NWM * NWN, BM = 32 * WM * NWM, BN = 32 * WN * NWN;
    constexpr int STAGE_FLOATS = (BM + BN) * BK;
    constexpr int PIECES = STAGE_FLOATS / 256;                 // 1-KiB DMA instructions per stage
    constexpr int PPW = (PIECES + NW - 1) / NW;                // per wavefront
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; i++) for (int j = 0; j < WN; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    unsigned goff[PPW];
    for (int p = 0; p < PPW; p++) {
        const unsigned piece = wave * PPW + p, row = piece * 16 + (lane >> 2);
        goff[p] = ((blockIdx.x * 4099u + row) * 384u + 4u * (lane & 3)) & src_mask;
    }
    unsigned kofs = 0;
    auto issue = [&](int stage) {
        float* base = lds + stage * STAGE_FLOATS;
#pragma unroll
        for (int p = 0; p < PPW; p++) {
            const int piece = wave * PPW + p;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds(GP(src + ((goff[p] + kofs) & src_mask)), LP(base + piece * 256), 16, 0, 0);
        }
        kofs += BK;
        if (kofs >= 384) kofs = 0;
    };
    // fragment read addresses (bytes): conflict-free swizzle is layout business of the real kernel; here rows of 64 B with the
    // same XOR the real kernel uses
    unsigned a_rd[WM], w_rd[WN];
    for (int i = 0; i < WM; i++) { const int r = 32 * (WM * wm + i) + (lane & 31); a_rd[i] = r * 64u + 16u * (((lane >> 5) * 2) ^ ((r >> 2) & 3)); }
    for (int j = 0; j < WN; j++) { const int r = 32 * (WN * wn + j) + (lane & 31); w_rd[j] = BM * 64u + r * 64u + 16u * (((lane >> 5) * 2) ^ ((r >> 2) & 3)); }
    auto compute = [&](int stage) {
        const unsigned base = (unsigned)(size_t)LP(lds + stage * STAGE_FLOATS);
#pragma unroll
        for (int kb = 0; kb < BK / 16; kb++) {
            f32x4 av[WM][2], wv[WN][2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
#pragma unroll
                for (int i = 0; i < WM; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(av[i][c]) : "v"((base + a_rd[i] + kb * (BM + BN) * 64u) ^ (16u * c)));
#pragma unroll
                for (int j = 0; j < WN; j++) asm volatile("ds_read_b128 %0, %1" : "=v"(wv[j][c]) : "v"((base + w_rd[j] + kb * (BM + BN) * 64u) ^ (16u * c)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < WM; i++) for (int c = 0; c < 2; c++) asm volatile("" : "+v"(av[i][c]));
#pragma unroll
            for (int j = 0; j < WN; j++) for (int c = 0; c < 2; c++) asm volatile("" : "+v"(wv[j][c]));
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < WM; i++)
#pragma unroll
                    for (int j = 0; j < WN; j++)
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c][k], wv[j][c][k], acc[i][j], 0, 0, 0);
        }
    };
    for (int s = 0; s < STAGES - 1; s++) issue(s);
    int stage = 0;
    for (int kc = 0; kc < chunks; kc++) {
        // chunk kc landed: at most STAGES-2 younger stage fills may stay in flight
        if (STAGES == 2 || kc + 1 >= chunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (STAGES == 3) { if (PPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if (PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (PPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (PPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else { if (PPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (PPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (kc + STAGES - 1 < chunks) issue((stage + STAGES - 1) % STAGES);
        compute(stage);
        stage = (stage + 1) % STAGES;
    }
    float s = 0.f;
    for (int i = 0; i < WM; i++) for (int j = 0; j < WN; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int WM, int WN, int NWM, int NWN, int BK, int STAGES, int MINBLK>
void run(const float* src, unsigned mask, float* out, int k_total) {
    constexpr int NW = NWM * NWN, BM = 32 * WM * NWM, BN = 32 * WN * NWN;
    const int lds_bytes = (BM + BN) * BK * 4 * STAGES;
    auto kern = skel<WM, WN, NWM, NWN, BK, STAGES, MINBLK>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * NW, lds_bytes);
    const int blocks = 256 * per_cu, chunks = k_total / BK;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds_bytes, 0, src, mask, out, chunks); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds_bytes, 0, src, mask, out, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * NW * chunks * (WM * WN * (BK / 2)) * (2.0 * 32 * 32 * 2);
    printf("WG %3dx%3d (%2d waves, wave tile %dx%d) BK %2d stages %d: %d WG/CU, LDS %3d KB, K=%5d: %.1f TFLOP/s  (%.3f ms)\n", BM, BN, NW, 32 * WM, 32 * WN, BK,
           STAGES, per_cu, lds_bytes / 1024, k_total, flops / (ms * 1e-3) / 1e12, ms / 5);
}
int main() {
    const size_t n = (size_t)1 << 28;
    float *src, *out; hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4); hipMalloc(&out, 1024 * 1024 * 4 * 4);
    const unsigned mask = (unsigned)(n - 1) & ~3u;
    for (int K : {384, 1536, 4608}) {
        run<2, 3, 2, 4, 16, 3, 1>(src, mask, out, K);
        run<2, 3, 2, 4, 16, 4, 1>(src, mask, out, K);
        run<2, 3, 4, 2, 16, 3, 1>(src, mask, out, K);
        run<2, 6, 2, 2, 16, 3, 1>(src, mask, out, K);
        run<4, 3, 1, 4, 16, 3, 1>(src, mask, out, K);
        run<2, 2, 2, 2, 16, 3, 3>(src, mask, out, K);
        run<2, 2, 2, 2, 16, 4, 2>(src, mask, out, K);
        run<2, 2, 2, 2, 32, 2, 2>(src, mask, out, K);
        run<2, 2, 4, 2, 16, 3, 2>(src, mask, out, K);
        run<2, 2, 4, 2, 16, 4, 1>(src, mask, out, K);
        run<2, 2, 4, 2, 32, 2, 1>(src, mask, out, K);
        run<2, 2, 4, 4, 16, 3, 1>(src, mask, out, K);
        run<2, 2, 4, 4, 16, 4, 1>(src, mask, out, K);
        run<2, 4, 2, 2, 16, 3, 2>(src, mask, out, K);
        run<2, 4, 4, 2, 16, 3, 1>(src, mask, out, K);
        run<4, 4, 2, 2, 16, 3, 1>(src, mask, out, K);
    }
    return 0;
}
