// self-corr-pose_amd/csrc/mutual_nn.hip -- row and column argmax of a masked score matrix in ONE pass.
//
// Replaces `pointcorr = pointcorr * (mask > 0) - 1e5 * (mask == 0); bw = pointcorr.max(1).indices;
// fw = pointcorr.max(2).indices` of PretrainedCorrespondence.match (model/module/pretrained_corr.py:85-89): the
// mutual nearest neighbours between the DINO key features of two images.  As torch ops this is six passes over the
// [N,P,Q] score tensor (outer-product mask, compare, select, two max reductions; 268 MB per pass at N=64, P=Q=1024).
// Here each workgroup owns a strip of 32 rows of one pair: every thread keeps the running (value, row) best of its 4
// columns in registers and the strip's row maxima are reduced across the workgroup; column partials of the strips are
// merged with a 64-bit atomicMax on (order-preserving float bits << 32 | ~row), row results are written directly.
// Ties resolve to the LOWEST index in both directions (the CPU semantics of torch.max).  HBM-bound: one read of S.
//
// scp_mutual_nn_fused (round 4): the score matrix itself is never formed.  pretrained_corr.py:85-89 is `bmm` -> mask -> max(1) /
// max(2): here one kernel per launch computes, for every (src image, tgt image) pair, 128 x 128 tiles of  S = K_src K_tgt^T
// (K = the DINO key features of the pair's images, token-major [tokens, 384]) on the matrix cores -- default: the split main loop
// of csrc/gemm_core_split.h (fp32 products as six bf16 MFMA products, fp32 accumulation), A operand = the fp32 keys split in
// registers, W operand = their three bf16 planes -- and reduces each tile IN REGISTERS to its masked row maxima / column maxima
// with the lowest index on ties, merged across tiles by 64-bit atomicMax on (order-preserving float bits << 32 | ~index).
// 51.5 GFLOP per step at B = 32 and no 268 MB score tensor ("0 if fused with argmax").
#include <hip/hip_runtime.h>

#include "gemm_core.h"
#include "gemm_core_split.h"
#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr int MN_THREADS = 256;
constexpr int MN_ROWS = 32;

__device__ __forceinline__ unsigned ordered(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(MN_THREADS) void mutual_argmax_kernel(const float* __restrict__ S,
                                                                   const float* __restrict__ rowmask,
                                                                   const float* __restrict__ colmask, int P, int Q,
                                                                   unsigned long long* __restrict__ colbest,
                                                                   long long* __restrict__ row_index) {
    __shared__ unsigned long long wave_best[MN_THREADS / 64];
    const int n = blockIdx.y;
    const int r0 = blockIdx.x * MN_ROWS, r1 = min(r0 + MN_ROWS, P);
    const float* Sn = S + (size_t)n * P * Q;
    for (int c0 = 0; c0 < Q; c0 += 4 * MN_THREADS) {            // column chunks of 1024 (one for the 256^2 config)
        const int c = c0 + 4 * threadIdx.x;
        const bool live = c < Q;
        float4 cm = {0, 0, 0, 0};
        if (live && colmask) cm = *reinterpret_cast<const float4*>(colmask + (size_t)n * Q + c);
        const bool k0 = !colmask || cm.x > 0, k1 = !colmask || cm.y > 0, k2 = !colmask || cm.z > 0, k3 = !colmask || cm.w > 0;
        float bv0 = -INFINITY, bv1 = -INFINITY, bv2 = -INFINITY, bv3 = -INFINITY;
        int br0 = 0, br1 = 0, br2 = 0, br3 = 0;
        for (int r = r0; r < r1; r++) {
            const bool rk = !rowmask || rowmask[(size_t)n * P + r] > 0;
            float4 v = {-1e5f, -1e5f, -1e5f, -1e5f};
            if (live) {
                const float4 s = *reinterpret_cast<const float4*>(Sn + (size_t)r * Q + c);
                if (rk && k0) v.x = s.x;
                if (rk && k1) v.y = s.y;
                if (rk && k2) v.z = s.z;
                if (rk && k3) v.w = s.w;
            }
            // columns: strictly greater keeps the earliest row
            if (v.x > bv0) { bv0 = v.x; br0 = r; }
            if (v.y > bv1) { bv1 = v.y; br1 = r; }
            if (v.z > bv2) { bv2 = v.z; br2 = r; }
            if (v.w > bv3) { bv3 = v.w; br3 = r; }
            // row: best of this thread's 4 columns (lowest column on ties), then across the workgroup
            float rv = v.x; int rc = c;
            if (v.y > rv) { rv = v.y; rc = c + 1; }
            if (v.z > rv) { rv = v.z; rc = c + 2; }
            if (v.w > rv) { rv = v.w; rc = c + 3; }
            unsigned long long key = live ? (((unsigned long long)ordered(rv) << 32) | (unsigned)(~(unsigned)rc)) : 0ull;
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(key, off, 64);
                key = o > key ? o : key;
            }
            if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = key;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long b = wave_best[0];
                for (int w = 1; w < MN_THREADS / 64; w++) b = wave_best[w] > b ? wave_best[w] : b;
                // chunks of one row are visited in increasing column order: keep the earlier chunk on ties
                unsigned long long* slot = reinterpret_cast<unsigned long long*>(row_index) + (size_t)n * P + r;
                if (c0 == 0 || b > *slot) *slot = b;
            }
            __syncthreads();
        }
        if (live) {
            unsigned long long* cb = colbest + (size_t)n * Q + c;
            atomicMax(cb + 0, ((unsigned long long)ordered(bv0) << 32) | (unsigned)(~(unsigned)br0));
            atomicMax(cb + 1, ((unsigned long long)ordered(bv1) << 32) | (unsigned)(~(unsigned)br1));
            atomicMax(cb + 2, ((unsigned long long)ordered(bv2) << 32) | (unsigned)(~(unsigned)br2));
            atomicMax(cb + 3, ((unsigned long long)ordered(bv3) << 32) | (unsigned)(~(unsigned)br3));
        }
    }
}

// packed keys -> indices (in place for the row slots, which were written as keys; out of place for the columns)
__global__ void mutual_unpack_kernel(const unsigned long long* __restrict__ colbest, long total_cols,
                                     long long* __restrict__ col_index, long long* __restrict__ row_index, long total_rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total_cols) col_index[i] = (long long)(unsigned)(~(unsigned)(colbest[i] & 0xFFFFFFFFull));
    if (i < total_rows) {
        const unsigned long long k = reinterpret_cast<unsigned long long*>(row_index)[i];
        row_index[i] = (long long)(unsigned)(~(unsigned)(k & 0xFFFFFFFFull));
    }
}


// ---- fused score GEMM + dual argmax -------------------------------------------------------------------------------------------
struct FusedArgs {
    const float* keys;            // [rows_total, C] fp32, token-major (A operand)
    const void* w;                // W operand: TILED planes of the keys (split core; rows padded to 32) or the same fp32 keys (fp32 core)
    const int* src_img;           // [N] image of the pair's source / target side
    const int* tgt_img;
    const float* mask;            // [B, P] per-image mask at the key resolution (> 0 = inside), or nullptr
    unsigned long long* rowbest;  // [N, P] packed (value, ~tgt index): best target per source token
    unsigned long long* colbest;  // [N, P] packed (value, ~src index): best source per target token
    int N, P, C, n_tok, tok0, rows_total;
};

// 128 x 128 tile, 4 wavefronts of 64 x 64, 40 KiB ring; A = fp32 keys split in registers, W = their TILED bf16 planes
// (csrc/gemm_core_split.h: every LDS-DMA piece one contiguous KiB), as the ViT's K projection leaves them or scp_split_bf16x3_tiled makes them
using FusedSplitCfg = scp::SplitCfg<2, 2, 2, 2, 2, 3, true, false, true>;
using FusedFp32Cfg = scp::GemmCfg<2, 2, 2, 2, 2, 2>;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// maximum over the 32 lanes of a half-wavefront, in every lane of it
__device__ __forceinline__ float half_max(float x) {
    x = fmaxf(x, dpp_f<0xB1>(x));        // lane ^ 1
    x = fmaxf(x, dpp_f<0x4E>(x));        // lane ^ 2
    x = fmaxf(x, dpp_f<0x141>(x));       // row_half_mirror: lane ^ 7
    x = fmaxf(x, dpp_f<0x140>(x));       // row_mirror: lane ^ 15
    return fmaxf(x, __shfl_xor(x, 16, 64));
}
__device__ __forceinline__ unsigned long long pack_key(float v, int idx) {
    return ((unsigned long long)ordered(v) << 32) | (unsigned)(~(unsigned)idx);
}

template <class CFG, class Core>
__global__ __launch_bounds__(CFG::THREADS, 2) void mutual_nn_fused_kernel(const FusedArgs g) {
    static_assert(CFG::WM == 2 && CFG::WN == 2, "the epilogue's lane <-> row / column slots assume 64 x 64 per wavefront");
    if constexpr (!std::is_same_v<Core, scp::GemmCore<CFG>>) scp::claim_vgprs<168>();      // split core = bf16 MFMAs (scp_common.h)
    __shared__ __attribute__((aligned(16))) float lds[CFG::LDS_BYTES / 4];
    // all tiles of a pair on one XCD (workgroup b runs on XCD b % 8): its 8 + 8 operand panels (3.9 MB) stay in that XCD's L2
    const int tiles_1d = (g.P + CFG::BM - 1) / CFG::BM, tiles = tiles_1d * tiles_1d;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int pair = (local / tiles) * 8 + xcd, tile = local % tiles;
    if (pair >= g.N) return;
    const int m0 = (tile / tiles_1d) * CFG::BM, n0 = (tile % tiles_1d) * CFG::BN;
    const int simg = g.src_img[pair], timg = g.tgt_img[pair];
    const int srow = simg * g.n_tok + g.tok0, trow = timg * g.n_tok + g.tok0;
    Core core(lds);
    core.set_rows(g.keys, g.w, g.rows_total, g.C, [&](int r) { return srow + min(m0 + r, g.P - 1); },
                  [&](int r) { return trow + min(n0 + r, g.P - 1); });
    typename Core::Acc acc;
    core.run(acc, g.C / CFG::BK);

    // ---- epilogue: a lane holds S[m][n] for column n = nb + 32 j + l31 (j = 0, 1) and the 32 rows mb + 32 i + acc_row(r, half)
    const int half = core.lane >> 5, l31 = core.lane & 31;
    const int mb = m0 + core.row_base(), nb = n0 + core.col_base();
    const float* smask = g.mask ? g.mask + (size_t)simg * g.P : nullptr;
    const float* tmask = g.mask ? g.mask + (size_t)timg * g.P : nullptr;
    bool ck[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = nb + 32 * j + l31;
        ck[j] = n < g.P && (!tmask || tmask[min(n, g.P - 1)] > 0);
    }
    float cv[2] = {-INFINITY, -INFINITY};
    int cr[2] = {0, 0};
    unsigned long long rowkey = 0ull;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = mb + 32 * i + scp::acc_row(r, half);
            const bool m_in = m < g.P;
            const bool rk = m_in && (!smask || smask[min(m, g.P - 1)] > 0);
            // masked entries count as -1e5 (pretrained_corr.py:86); rows / columns past the matrix never win
            const float v0 = (nb + l31 < g.P) ? ((rk && ck[0]) ? acc.t[2 * i][r] : -1e5f) : -INFINITY;
            const float v1 = (nb + 32 + l31 < g.P) ? ((rk && ck[1]) ? acc.t[2 * i + 1][r] : -1e5f) : -INFINITY;
            // columns: rows are visited in increasing order, strictly greater keeps the lowest row
            if (m_in && v0 > cv[0]) { cv[0] = v0; cr[0] = m; }
            if (m_in && v1 > cv[1]) { cv[1] = v1; cr[1] = m; }
            // row m: maximum over this wavefront's 64 columns, then the LOWEST column that attains it
            const float mx = half_max(fmaxf(v0, v1));
            const unsigned long long b0 = __ballot(v0 == mx), b1 = __ballot(v1 == mx);
            const unsigned h0 = (unsigned)(b0 >> (32 * half)), h1 = (unsigned)(b1 >> (32 * half));
            const int col = h0 ? __builtin_ctz(h0) : 32 + __builtin_ctz(h1 | 0x80000000u);
            if (l31 == 16 * i + r) rowkey = m_in ? pack_key(mx, nb + col) : 0ull;
        }
    }
    // row slots: lane l31 of half h owns row mb + 32 (l31 >> 4) + acc_row(l31 & 15, h)
    {
        const int m = mb + 32 * (l31 >> 4) + scp::acc_row(l31 & 15, half);
        if (m < g.P && rowkey) atomicMax(g.rowbest + (size_t)pair * g.P + m, rowkey);
    }
    // column slots: combine the two halves (they hold different rows of the same columns), then half h writes column block j = h
#pragma unroll
    for (int j = 0; j < 2; j++) {
        unsigned long long k = cv[j] > -INFINITY ? pack_key(cv[j], cr[j]) : 0ull;
        const unsigned long long o = __shfl_xor(k, 32, 64);
        k = o > k ? o : k;
        const int n = nb + 32 * j + l31;
        if (half == j && n < g.P && k) atomicMax(g.colbest + (size_t)pair * g.P + n, k);
    }
}

__global__ void unpack_keys_kernel(const unsigned long long* __restrict__ keys, long long* __restrict__ a, long long* __restrict__ b,
                                   long n_each) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n_each) return;
    const long long idx = (long long)(unsigned)(~(unsigned)(keys[i] & 0xFFFFFFFFull));
    if (i < n_each) a[i] = idx;
    else b[i - n_each] = idx;
}

}  // namespace

extern "C" size_t scp_mutual_argmax_workspace(int N, int Q) { return (size_t)N * Q * sizeof(unsigned long long); }

extern "C" int scp_mutual_argmax(const float* scores, const float* rowmask, const float* colmask, int N, int P, int Q,
                                 long long* col_index, long long* row_index, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    if (N <= 0 || P <= 0 || Q <= 0) return scp::fail(hipErrorInvalidValue, "mutual_argmax: empty problem");
    if (Q % 4 != 0) return scp::fail(hipErrorInvalidValue, "mutual_argmax: Q must be a multiple of 4");
    if (!scores || !col_index || !row_index) return scp::fail(hipErrorInvalidValue, "mutual_argmax: null argument");
    if (!workspace || workspace_bytes < scp_mutual_argmax_workspace(N, Q))
        return scp::fail(hipErrorInvalidValue, "mutual_argmax: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long* colbest = static_cast<unsigned long long*>(workspace);
    if (hipMemsetAsync(colbest, 0, (size_t)N * Q * sizeof(unsigned long long), st) != hipSuccess)
        return scp::check_launch("mutual_argmax memset");
    hipLaunchKernelGGL(mutual_argmax_kernel, dim3((P + MN_ROWS - 1) / MN_ROWS, N), dim3(MN_THREADS), 0, st, scores, rowmask,
                       colmask, P, Q, colbest, row_index);
    if (int e = scp::check_launch("mutual_argmax")) return e;
    const long tc = (long)N * Q, tr = (long)N * P;
    const long total = tc > tr ? tc : tr;
    hipLaunchKernelGGL(mutual_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, colbest, tc, col_index,
                       row_index, tr);
    return scp::check_launch("mutual_argmax unpack");
}

extern "C" size_t scp_mutual_nn_fused_workspace(int N, int P) { return 2 * (size_t)N * P * sizeof(unsigned long long); }

extern "C" int scp_mutual_nn_fused(const float* keys, const void* key_planes, int n_images, int n_tok, int tok0, int C,
                                   const int* src_img, const int* tgt_img, const float* mask, int N, int P, long long* tgt_of_src,
                                   long long* src_of_tgt, void* workspace, size_t workspace_bytes, void* stream) {
    if (N <= 0 || P <= 0 || n_images <= 0) return scp::fail(hipErrorInvalidValue, "mutual_nn_fused: empty problem");
    if (!keys || !src_img || !tgt_img || !tgt_of_src || !src_of_tgt) return scp::fail(hipErrorInvalidValue, "mutual_nn_fused: null argument");
    if (C <= 0 || C % 32 != 0) return scp::fail(hipErrorInvalidValue, "mutual_nn_fused: C must be a multiple of 32");
    if (tok0 < 0 || tok0 + P > n_tok) return scp::fail(hipErrorInvalidValue, "mutual_nn_fused: tok0 + P exceeds the tokens per image");
    if ((size_t)n_images * n_tok * C >= (1ull << 30)) return scp::fail(hipErrorInvalidValue, "mutual_nn_fused: keys larger than 2^30 elements");
    if (!workspace || workspace_bytes < scp_mutual_nn_fused_workspace(N, P))
        return scp::fail(hipErrorInvalidValue, "mutual_nn_fused: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long* best = static_cast<unsigned long long*>(workspace);
    if (hipMemsetAsync(best, 0, scp_mutual_nn_fused_workspace(N, P), st) != hipSuccess) return scp::check_launch("mutual_nn_fused memset");
    FusedArgs g{};
    g.keys = keys; g.w = key_planes ? key_planes : static_cast<const void*>(keys);
    g.src_img = src_img; g.tgt_img = tgt_img; g.mask = mask;
    g.rowbest = best; g.colbest = best + (size_t)N * P;
    g.N = N; g.P = P; g.C = C; g.n_tok = n_tok; g.tok0 = tok0; g.rows_total = n_images * n_tok;
    const int tiles_1d = (P + FusedSplitCfg::BM - 1) / FusedSplitCfg::BM;
    const unsigned grid = (unsigned)(((N + 7) / 8) * tiles_1d * tiles_1d * 8);
    if (key_planes)
        hipLaunchKernelGGL((mutual_nn_fused_kernel<FusedSplitCfg, scp::SplitGemmCore<FusedSplitCfg>>), dim3(grid), dim3(FusedSplitCfg::THREADS),
                           0, st, g);
    else
        hipLaunchKernelGGL((mutual_nn_fused_kernel<FusedFp32Cfg, scp::GemmCore<FusedFp32Cfg>>), dim3(grid), dim3(FusedFp32Cfg::THREADS), 0, st, g);
    if (int e = scp::check_launch("mutual_nn_fused")) return e;
    const long n_each = (long)N * P;
    hipLaunchKernelGGL(unpack_keys_kernel, dim3((unsigned)((2 * n_each + 255) / 256)), dim3(256), 0, st, best, tgt_of_src, src_of_tgt, n_each);
    return scp::check_launch("mutual_nn_fused unpack");
}
