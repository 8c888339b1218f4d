// self-corr-pose_amd/csrc/vit_gemm.hip -- the linear layers of the frozen DINO ViT-S/8 on the gfx950 fp32 matrix cores,
// with everything around them fused into the GEMM.
//
// Replaces, per transformer block (third-party/zsp/zsp/method/vision_transformer_flexible.py):
//   :126-132  x = x + attn(norm1(x)); x = x + mlp(norm2(x))       Block.forward
//   :85-101   qkv = Linear(dim, 3 dim)(.), proj = Linear(dim, dim)  Attention
//   :54-70    fc2(GELU(fc1(.)))                                     Mlp (nn.GELU = erf form)
// i.e. four GEMMs  C[M,N] = A[M,K] W[N,K]^T  (M = B*1025 tokens, K, N in {384, 1152, 1536}) plus two LayerNorms, a GELU,
// two bias+residual adds -- eight extra passes over [M,384]...[M,1536] activations when run as separate kernels.
//
// Fusions (one kernel family, epilogue selected at compile time):
//   LN prologue, folded algebraically: LayerNorm(x) W^T = rstd_m * (x (gamma o W)^T)_mn - rstd_m mu_m s_n + t_n with
//       s_n = sum_k gamma_k W_nk,  t_n = sum_k beta_k W_nk + bias_n.  The weights are frozen, so gamma o W, s, t are built
//       once; the GEMM streams the RAW residual stream x and applies (mu_m, rstd_m) -- one tiny row-statistics kernel per
//       LayerNorm -- in its epilogue.  The normalised activation is never written or read.
//   EPI_LN            qkv  = LN1(x) Wqkv^T + b
//   EPI_LN_GELU       h    = GELU(LN2(x) W1^T + b1)
//   EPI_BIAS_RESIDUAL x   += y W^T + b      (proj and fc2; in place on the residual stream)
//   EPI_BIAS          plain Linear (block 9's K slice uses EPI_LN with a 384-row weight slice)
//
// CDNA4 mapping: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).  Workgroup = 4 wavefronts = 128 x 128
// output tile, wavefront = 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs).  Both operands are K-contiguous, so a lane
// (row = lane & 31, half = lane >> 5) takes its 8 k-values of a 16-wide K chunk with two ds_read_b128 and feeds them to
// 8 MFMAs unchanged (A and W use the same k <-> (half, register) assignment; any pairing is a valid contraction order).
// Tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs), a three-stage ring of SEPARATE LDS objects, one
// barrier per K chunk; the DMA destination is lane-linear, so bank conflicts are removed by permuting the SOURCE address:
// 16-byte chunk c of tile row r lands in slot c ^ ((r >> 2) & 3) and is read back through the same XOR (16 consecutive rows
// hit 16 distinct 4-bank groups).  48 KiB of LDS (three stages) and <= 168 VGPRs per workgroup -> 3 workgroups per CU = 768 slots: at
// B = 32 the 256 full row panels x {3, 9, 12} column blocks are exact multiples of 768 (no partial last round).
// blockIdx is remapped so that the N-blocks of one 128-row panel of A run back to back on ONE XCD: A is fetched from HBM
// once, W (<= 2.4 MB) lives in every XCD's L2.
// Roofline: bound = fp32 MFMA (157.3 TFLOP/s); algorithmic flops 2 M N K per launch; algorithmic bytes 4 (M K + N K + M N).
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int THREADS = 256;
constexpr int TILE_FLOATS = BM * BK;            // one operand tile of one stage (8 KiB)

#define SCP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SCP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct GemmArgs {
    const float* A;        // [M, K]
    const float* W;        // [N, K]
    const float* vec0;     // EPI_LN*: s[N]            EPI_BIAS*: bias[N]
    const float* vec1;     // EPI_LN*: t[N]
    const float* rowstat;  // EPI_LN*: (mean, rstd)[M]
    const float* resid;    // EPI_BIAS_RESIDUAL: [M, N] (may alias C)
    float* C;              // [M, N]
    int M, N, K;
    int nblk_n, full_panels, rem_blocks, per_xcd;
    const int* m_dev;      // optional: the row count lives on the device (<= M); the tile bookkeeping is then redone in the kernel
    const int* a_rows;     // optional: GEMM row m reads A row a_rows[m] ...
    const int* c_rows;     // ... and its rowstat / resid / C row is c_rows[m] (row gather / scatter without a copy)
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int EPI, bool INDEXED>
__global__ __launch_bounds__(THREADS, 3) void vit_gemm_kernel(const GemmArgs g) {
    // one LDS object per (operand, stage): the compiler's wait-count insertion tracks LDS-DMA writes per object, so a
    // ds_read of stage 0 does not have to wait for the DMA that is filling stage 1 (with a_lds[2][..] it inserted
    // s_waitcnt vmcnt(0) in front of every read and serialised prefetch and compute)
    __shared__ __attribute__((aligned(16))) float a_lds0[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float a_lds1[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float a_lds2[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float w_lds0[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float w_lds1[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float w_lds2[TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int prow = lane >> 2, pslot = lane & 3;

    // Tile order.  (1) Full 128-row panels in an XCD-aware order: workgroup b runs on XCD b % 8, consecutive logical ids of
    // one XCD (adjacent in time) walk the N-blocks of one A panel, and every XCD gets the same number of full tiles.
    // (2) The N-blocks of the SHORT last panel (M = B * 1025 tokens leaves 32 rows) come last and spread their one 32-row
    // strip over the four wavefronts (32 x 32 each).  Every workgroup walks tiles t = blockIdx.x, + gridDim.x, ..., so any grid
    // size is correct; the launcher uses one workgroup per tile (see `launch`).  When a workgroup does run several tiles, the
    // LDS-DMA prologue of the next tile is issued before the epilogue of the current one.
    // row count: host value, or read from the device (rows selected by an earlier kernel, no host round trip)
    int M = g.M, full_panels = g.full_panels, rem_blocks = g.rem_blocks, per_xcd = g.per_xcd;
    if (g.m_dev) {
        M = min(max(__builtin_amdgcn_readfirstlane(*g.m_dev), 0), g.M);
        const int tail_rows = M % BM;
        full_panels = M / BM + (tail_rows > 32 ? 1 : 0);
        rem_blocks = (tail_rows > 0 && tail_rows <= 32) ? g.nblk_n : 0;
        per_xcd = (full_panels * g.nblk_n + 7) / 8;
    }
    const int full_slots = per_xcd * 8, total = full_slots + rem_blocks;
    struct Tile { int m0, n0; bool rem, ok; };
    auto tile_of = [&](int t) {
        Tile x;
        x.rem = t >= full_slots;
        int bm, bn;
        if (x.rem) {
            bn = t - full_slots;
            bm = full_panels;
            x.ok = true;
        } else {
            const int lid = (t & 7) * per_xcd + (t >> 3);
            x.ok = lid < full_panels * g.nblk_n;
            bm = lid / g.nblk_n;
            bn = lid - bm * g.nblk_n;
        }
        x.m0 = bm * BM;
        x.n0 = bn * BN;
        return x;
    };
    // LDS-DMA pieces.  An operand tile of one stage is 128 rows x 64 B = 8 instructions of 1 KiB (16 rows each); the 16
    // pieces of (A tile, W tile) are dealt to the 4 wavefronts: wavefront w moves A pieces 2w, 2w+1 and W pieces alike.
    // Per piece the per-lane part of the source address (row, swizzled chunk) is loop invariant within a tile.
    unsigned a_off[2], w_off[2];
    auto set_offsets = [&](const Tile& x) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = 16 * (2 * wave + i) + prow;                      // tile row 0..127
            const int chunk = pslot ^ ((r >> 2) & 3);
            int arow = min(x.m0 + r, M - 1);
            if (INDEXED && g.a_rows) arow = g.a_rows[arow];
            a_off[i] = (unsigned)arow * (unsigned)g.K + 4u * chunk;
            w_off[i] = (unsigned)min(x.n0 + r, g.N - 1) * (unsigned)g.K + 4u * chunk;
        }
    };
    auto issue_stage = [&](int kc, float* a_dst, float* w_dst) {
        const float* ap = g.A + kc * BK;
        const float* wp = g.W + kc * BK;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(ap + a_off[i]), SCP_LDS_PTR(a_dst + (2 * wave + i) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(wp + w_off[i]), SCP_LDS_PTR(w_dst + (2 * wave + i) * 256), 16, 0, 0);
        }
    };
    const int nk = g.K / BK;

    int t = blockIdx.x;
    Tile cur = tile_of(t);
    while (t < total && !cur.ok) { t += gridDim.x; if (t < total) cur = tile_of(t); }
    if (t >= total) return;
    set_offsets(cur);
    issue_stage(0, a_lds0, w_lds0);
    if (nk > 1) issue_stage(1, a_lds1, w_lds1);

    while (true) {
        // wavefront's sub-tile: rows row_base + 32 i, columns col_base + 32 j.  Full panel: 64 x 64 (2 x 2 MFMA tiles);
        // remainder panel: the strip's 32 rows x 32 columns per wavefront (one MFMA tile)
        const bool rem = cur.rem;
        const int row_base = rem ? 0 : 64 * (wave >> 1);
        const int col_base = rem ? 32 * wave : 64 * (wave & 1);
#if defined(SCP_GEMM_ABLATE) && (SCP_GEMM_ABLATE & 4)
        const bool tile_live[2] = {true, true};
#else
        const bool tile_live[2] = {true, !rem};      // [i] and [j] alike
#endif
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        // lane's read offsets (floats) inside a stage: rows row_base + 32*i + l31 of A, col_base + 32*j + l31 of W; chunk 2*half + c
        int a_rd[2][2], w_rd[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int ra = row_base + 32 * i + l31, rw = col_base + 32 * i + l31;
                a_rd[i][c] = ra * BK + 4 * ((2 * half + c) ^ ((ra >> 2) & 3));
                w_rd[i][c] = rw * BK + 4 * ((2 * half + c) ^ ((rw >> 2) & 3));
            }
        // LDS reads of the operand fragments are written as ds_read_b128 instructions by hand: for a compiler-visible LDS load
        // the wait-count pass assumes it may alias the LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of it, which
        // serialises the two-chunk prefetch.  The explicit lgkmcnt(0) below carries the fragments as operands, so that no
        // MFMA is scheduled above it.
        auto compute_stage = [&](const float* as, const float* ws) {
            const unsigned a_base = (unsigned)(size_t)SCP_LDS_PTR(as), w_base = (unsigned)(size_t)SCP_LDS_PTR(ws);
            f32x4 av[2][2], wv[2][2];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(av[i][c]) : "v"(a_base + 4u * (unsigned)a_rd[i][c]));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(wv[i][c]) : "v"(w_base + 4u * (unsigned)w_rd[i][c]));
                }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(wv[0][0]), "+v"(wv[0][1]),
                           "+v"(wv[1][0]), "+v"(wv[1][1]));
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    if (!tile_live[i]) continue;                       // wavefront-uniform
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if (!tile_live[j]) continue;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].x, wv[j][c].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].y, wv[j][c].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].z, wv[j][c].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].w, wv[j][c].w, acc[i][j], 0, 0, 0);
                    }
                }
        };
        // three-stage ring, prefetch distance two chunks: while chunk kc is multiplied, kc+1 has been in flight for a whole
        // chunk time and kc+2 is issued.  Each stage issue is 4 DMA instructions per wavefront, so "chunk kc has landed" is
        // vmcnt(4) while a younger chunk is outstanding and vmcnt(0) at the tail.  Chunks 0 and 1 were issued by the prologue
        // (of the kernel, or of the previous tile's epilogue phase).
        auto step = [&](int kc, const float* as, const float* ws, float* a_next, float* w_next) {
            if (kc + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // chunk kc visible to all; every wavefront is done with chunk kc-1, whose stage is refilled now.  A bare s_barrier:
            // __syncthreads() carries a workgroup release fence, for which the compiler waits for ALL outstanding LDS-DMA
            // (vmcnt(0)) -- that would cut the prefetch distance from two chunks to one.  The LDS reads of chunk kc-1 were
            // consumed by its MFMAs (lgkmcnt(0)), so nothing of this wavefront is in flight on the stage that is refilled.
            __builtin_amdgcn_s_barrier();
            if (kc + 2 < nk) issue_stage(kc + 2, a_next, w_next);
            compute_stage(as, ws);
        };
        for (int kc = 0; kc < nk; kc += 3) {
            step(kc, a_lds0, w_lds0, a_lds2, w_lds2);
            if (kc + 1 < nk) step(kc + 1, a_lds1, w_lds1, a_lds0, w_lds0);
            if (kc + 2 < nk) step(kc + 2, a_lds2, w_lds2, a_lds1, w_lds1);
        }

        // ---- next tile: start its first two chunks before this tile's epilogue (every wavefront is done with the LDS ring)
        int tn = t + gridDim.x;
        Tile nxt = cur;
        bool has_next = false;
        while (tn < total) {
            nxt = tile_of(tn);
            if (nxt.ok) { has_next = true; break; }
            tn += gridDim.x;
        }
        __syncthreads();
        if (has_next) {
            set_offsets(nxt);
            // the ring position of the next tile's chunk 0 / 1 must be stages 0 / 1 again
            issue_stage(0, a_lds0, w_lds0);
            if (nk > 1) issue_stage(1, a_lds1, w_lds1);
        }

        // ---- epilogue.  MFMA layout: A operand rows -> accumulator rows acc_row(reg, half), B operand rows (W rows = output
        // columns) -> lane & 31: lane holds C[m][n = n_base + l31] for 16 rows m -> 32 consecutive floats per row per half-wave.
        // All loads of a 32 x 32 tile are issued before its stores (resid may alias C element for element; every element is read
        // and written by the same lane only).
        const int m0 = cur.m0, n0 = cur.n0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (!tile_live[i]) continue;
            const int mb = m0 + row_base + 32 * i;
            // output-side row (rowstat / resid / C) of accumulator row r; looked up at each use (an L1 hit) rather than held
            // in 16 registers through the epilogue
            auto orow = [&](int r) {
                const int m = min(mb + acc_row(r, half), M - 1);
                return (INDEXED && g.c_rows) ? g.c_rows[m] : m;
            };
            float mean[16], rstd[16];
            if (EPI == SCP_GEMM_LN || EPI == SCP_GEMM_LN_GELU) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = orow(r);
                    const float2 st = *reinterpret_cast<const float2*>(g.rowstat + 2 * (size_t)m);
                    mean[r] = st.x;
                    rstd[r] = st.y;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (!tile_live[j]) continue;
                const int n = n0 + col_base + 32 * j + l31;
                const bool n_ok = n < g.N;
                const int nc = min(n, g.N - 1);
                const float v0 = g.vec0[nc];
                float v1 = 0.f;
                if (EPI == SCP_GEMM_LN || EPI == SCP_GEMM_LN_GELU) v1 = g.vec1[nc];
                float res[16];
                if (EPI == SCP_GEMM_BIAS_RESIDUAL) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        res[r] = g.resid[(size_t)orow(r) * g.N + nc];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + acc_row(r, half);
                    float x = acc[i][j][r];
                    if (EPI == SCP_GEMM_LN || EPI == SCP_GEMM_LN_GELU) {
                        x = rstd[r] * (x - mean[r] * v0) + v1;
                        if (EPI == SCP_GEMM_LN_GELU) x = gelu_erf(x);
                    } else {
                        x += v0;
                        if (EPI == SCP_GEMM_BIAS_RESIDUAL) x += res[r];
                    }
#if defined(SCP_GEMM_ABLATE) && (SCP_GEMM_ABLATE & 1)
                    if (x == 12345.678f)                       // timing ablation only (tools/probes): no output traffic
#endif
                    if (m < M && n_ok) g.C[(size_t)orow(r) * g.N + n] = x;
                }
            }
        }
        if (!has_next) break;
        t = tn;
        cur = nxt;
    }
}

// per-row LayerNorm statistics (mean, rstd = 1 / sqrt(var + eps)), biased variance as nn.LayerNorm, two-pass over registers.
// One wavefront handles 4 rows at a time with all of their loads in flight (C <= 1536, C % 4 == 0: <= 6 float4 per lane).
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, int rows,
                                                        int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= rows) return;
    const int nq = C >> 2;
    float4 v[4][6];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)min(row0 + r, rows - 1) * C);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int q = lane + 64 * i;
            v[r][i] = q < nq ? xr[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        const float mean = s / C;
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (lane + 64 * i < nq) {
                const float dx = v[r][i].x - mean, dy = v[r][i].y - mean, dz = v[r][i].z - mean, dw = v[r][i].w - mean;
                q2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) q2 += __shfl_xor(q2, m);
        if (lane == 0 && row0 + r < rows) {
            stats[2 * (size_t)(row0 + r)] = mean;
            stats[2 * (size_t)(row0 + r) + 1] = 1.f / sqrtf(q2 / C + eps);
        }
    }
}

// C = 384 (every LayerNorm of ViT-S): 16 lanes per row (6 float4 each, 256 B contiguous per load instruction), two rows per
// thread in flight, reductions inside the 16-lane DPP row (quad_perm / row_half_mirror / row_mirror: no LDS, no ds_bpermute
// chains -- the generic kernel below spends most of its 50 us in 48 dependent cross-lane steps per wavefront).
template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_get<0xB1>(x);      // lane ^ 1
    x += dpp_get<0x4E>(x);      // lane ^ 2
    x += dpp_get<0x141>(x);     // lane ^ 7 (row_half_mirror)
    x += dpp_get<0x140>(x);     // lane ^ 15 (row_mirror)
    return x;
}
__global__ __launch_bounds__(256) void row_stats384_kernel(const float* __restrict__ x, float* __restrict__ stats, int rows,
                                                           float eps) {
    constexpr int C = 384;
    const int sub = threadIdx.x & 15;
    const int row0 = blockIdx.x * 32 + (threadIdx.x >> 4);          // rows row0 and row0 + 16
    float4 v[2][6];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)min(row0 + 16 * r, rows - 1) * C);
#pragma unroll
        for (int i = 0; i < 6; i++) v[r][i] = xr[sub + 16 * i];
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = row16_sum(s) / C;
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const float dx = v[r][i].x - mean, dy = v[r][i].y - mean, dz = v[r][i].z - mean, dw = v[r][i].w - mean;
            q2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        q2 = row16_sum(q2);
        const int row = row0 + 16 * r;
        if (sub == 0 && row < rows) *reinterpret_cast<float2*>(stats + 2 * (size_t)row) = make_float2(mean, 1.f / sqrtf(q2 / C + eps));
    }
}

template <int EPI>
void launch(const GemmArgs& g, hipStream_t st) {
    const int total = g.per_xcd * 8 + g.rem_blocks;
    // One workgroup per tile.  The kernel's tile loop also works with fewer workgroups than tiles (a persistent grid of
    // resident_slots() workgroups is equally fast alone: 111 / 97 / 110 / 113 TFLOP/s either way), but inside the training step
    // the ViT shares the device with the encoder's streams, and a persistent grid holds every CU for the whole launch: the
    // other streams' kernels then only start between GEMMs.  With one workgroup per tile slots are released tile by tile and the
    // step is 0.4 ms shorter (40.8 -> 40.4 ms).
    const dim3 grid(max(total, 1));
    // the row-index variant is a separate instantiation: the plain one keeps its register allocation
    if (g.a_rows || g.c_rows) hipLaunchKernelGGL((vit_gemm_kernel<EPI, true>), grid, dim3(THREADS), 0, st, g);
    else hipLaunchKernelGGL((vit_gemm_kernel<EPI, false>), grid, dim3(THREADS), 0, st, g);
}

}  // namespace

namespace {
int vit_linear_impl(const float* A, const float* W, const float* vec0, const float* vec1, const float* rowstat, const float* resid,
                    float* C, int M, const int* m_dev, const int* a_rows, const int* c_rows, int N, int K, int epilogue, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return scp::fail(hipErrorInvalidValue, "vit_linear: empty problem");
    if (K % (2 * BK) != 0) return scp::fail(hipErrorInvalidValue, "vit_linear: K must be a multiple of 32");
    if ((size_t)M * (size_t)K >= (1ull << 32) || (size_t)N * (size_t)K >= (1ull << 32))
        return scp::fail(hipErrorInvalidValue, "vit_linear: operand larger than 2^32 elements");
    const bool ln = epilogue == SCP_GEMM_LN || epilogue == SCP_GEMM_LN_GELU;
    if (!vec0 || (ln && (!vec1 || !rowstat)) || (epilogue == SCP_GEMM_BIAS_RESIDUAL && !resid))
        return scp::fail(hipErrorInvalidValue, "vit_linear: missing epilogue operand");
    GemmArgs g{};
    g.A = A; g.W = W; g.vec0 = vec0; g.vec1 = vec1; g.rowstat = rowstat; g.resid = resid; g.C = C;
    g.M = M; g.N = N; g.K = K; g.m_dev = m_dev; g.a_rows = a_rows; g.c_rows = c_rows;
    g.nblk_n = (N + BN - 1) / BN;
    // a last panel of <= 32 rows (M = B * 1025 tokens at B = 32 k) runs in strip mode; a longer one is an ordinary panel with
    // clamped loads and masked stores
    const int tail_rows = M % BM;
    g.full_panels = M / BM + (tail_rows > 32 ? 1 : 0);
    g.rem_blocks = (tail_rows > 0 && tail_rows <= 32) ? g.nblk_n : 0;
    g.per_xcd = (g.full_panels * g.nblk_n + 7) / 8;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (epilogue) {
        case SCP_GEMM_BIAS: launch<SCP_GEMM_BIAS>(g, st); break;
        case SCP_GEMM_BIAS_RESIDUAL: launch<SCP_GEMM_BIAS_RESIDUAL>(g, st); break;
        case SCP_GEMM_LN: launch<SCP_GEMM_LN>(g, st); break;
        case SCP_GEMM_LN_GELU: launch<SCP_GEMM_LN_GELU>(g, st); break;
        default: return scp::fail(hipErrorInvalidValue, "vit_linear: unknown epilogue");
    }
    return scp::check_launch("vit_linear");
}
}  // namespace

extern "C" int scp_vit_linear(const float* A, const float* W, const float* vec0, const float* vec1, const float* rowstat,
                              const float* resid, float* C, int M, int N, int K, int epilogue, void* stream) {
    return vit_linear_impl(A, W, vec0, vec1, rowstat, resid, C, M, nullptr, nullptr, nullptr, N, K, epilogue, stream);
}

extern "C" int scp_vit_linear_rows(const float* A, const float* W, const float* vec0, const float* vec1, const float* rowstat,
                                   const float* resid, float* C, const int* rows_dev, int max_rows, const int* a_rows,
                                   const int* c_rows, int N, int K, int epilogue, void* stream) {
    if (!rows_dev) return scp::fail(hipErrorInvalidValue, "vit_linear_rows: null row count");
    return vit_linear_impl(A, W, vec0, vec1, rowstat, resid, C, max_rows, rows_dev, a_rows, c_rows, N, K, epilogue, stream);
}

extern "C" int scp_row_mean_rstd(const float* x, float* stats, int rows, int C, float eps, void* stream) {
    if (rows <= 0) return 0;
    if (C <= 0 || C > 1536 || (C & 3)) return scp::fail(hipErrorInvalidValue, "row_mean_rstd: C must be a multiple of 4 in 4..1536");
    if (C == 384)
        hipLaunchKernelGGL(row_stats384_kernel, dim3((rows + 31) / 32), dim3(256), 0, static_cast<hipStream_t>(stream), x, stats, rows,
                           eps);
    else
        hipLaunchKernelGGL(row_stats_kernel, dim3((rows + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), x, stats, rows,
                           C, eps);
    return scp::check_launch("row_mean_rstd");
}
