// tools/probes/vit_gemm_bigtile_experiment.hip -- NOT part of the library: csrc/vit_gemm.hip generalised to (NWM x NWN) wavefronts of
// (WM x WN) MFMA tiles.  256 x 192 (8 wavefronts, 1 workgroup per CU) and 128 x 192 (4 wavefronts, 2 per CU) were measured at
// 108 / 91 / 106 / 107 and 107 / 88 / 108 / 106 TFLOP/s on the four ViT layers -- below the shipped 128 x 128 x 3-per-CU kernel
// (111 / 97 / 110 / 113) although the bare loop skeleton of these shapes (gemm_shapes.hip) is 8-12 % faster: with one or two
// workgroups per CU the epilogue and the remainder strip are no longer hidden behind other workgroups.
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
// CDNA4 mapping: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).  Workgroup = 8 wavefronts (4 x 2) = 256 x 192
// output tile, wavefront = 64 x 96 = 2 x 3 MFMA tiles (96 accumulator VGPRs), ONE workgroup per CU (2 wavefronts per SIMD).
// The tile shape is what the operand feed allows: every K chunk of 16 moves (BM + BN) x 64 B through LDS-DMA into LDS and
// BM x BN x 16 MACs out of it; a 128 x 128 tile at 3 workgroups per CU reached 111 TFLOP/s, the same loop skeleton with this
// shape feeds 10 % more (tools/probes/gemm_shapes.hip), and 192 divides every N of the ViT (384, 1152, 1536) while 256 x {2, 6,
// 8} column blocks x 128 row panels are exact multiples of the 256 CUs.  Both operands are K-contiguous, so a lane
// (row = lane & 31, half = lane >> 5) takes its 8 k-values of a 16-wide K chunk with two ds_read_b128 and feeds them to
// 8 MFMAs unchanged (A and W use the same k <-> (half, register) assignment; any pairing is a valid contraction order).
// Tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs) in a three-stage ring (84 KiB), one barrier per K
// chunk; the DMA destination is lane-linear, so bank conflicts are removed by permuting the SOURCE address: 16-byte chunk c
// of tile row r lands in slot c ^ ((r >> 2) & 3) and is read back through the same XOR (the four 16-lane groups of a
// ds_read_b128 hit 16 distinct slots).  The kernel is persistent (grid = #CU); tile t runs on XCD t % 8 and the N-blocks of
// one 256-row panel of A run back to back on ONE XCD: A is fetched from HBM once, W (<= 2.4 MB) lives in every XCD's L2.
// Roofline: bound = fp32 MFMA (157.3 TFLOP/s); algorithmic flops 2 M N K per launch; algorithmic bytes 4 (M K + N K + M N).
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NWM = 2, NWN = 2;                  // wavefronts along M, N
constexpr int WM = 2, WN = 3;                    // 32 x 32 MFMA tiles per wavefront along M, N
constexpr int BM = 32 * WM * NWM, BN = 32 * WN * NWN, BK = 16;   // 256 x 192 x 16
constexpr int NWAVES = NWM * NWN, THREADS = 64 * NWAVES;
constexpr int A_PIECES = BM / 16, W_PIECES = BN / 16;            // 1-KiB LDS-DMA instructions (16 rows x 64 B) per stage
constexpr int STAGE_FLOATS = (BM + BN) * BK;     // A tile then W tile of one stage (28 KiB)
constexpr int W_BASE = BM * BK;
constexpr int STRIP_TN = (BN / 32 + NWAVES - 1) / NWAVES;      // remainder strip (32 rows x BN): MFMA tiles per wavefront ...
constexpr int STRIP_WAVES = (BN / 32 + STRIP_TN - 1) / STRIP_TN;  // ... and wavefronts that own some
constexpr int A_PER = A_PIECES / NWAVES, W_PER = (W_PIECES + NWAVES - 1) / NWAVES;   // pieces per wavefront per stage
static_assert(A_PIECES % NWAVES == 0, "A pieces are dealt evenly");
#ifndef SCP_GEMM_WG_PER_CU
#define SCP_GEMM_WG_PER_CU 2
#endif

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
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

struct Tile { int m0, n0; bool rem, ok; };

// LDS-DMA pieces.  A stage is 16 A pieces + 12 W pieces of 1 KiB (16 rows x 64 B each).  Wavefront w moves A pieces w and
// w + 8, W piece w, and (w < 4 only) W piece w + 8: 4 instructions per stage for wavefronts 0-3, 3 for wavefronts 4-7.
// Per piece the per-lane part of the source address (row, swizzled chunk) is loop invariant within a tile.
struct Feeder {
    const float *A, *W;
    int M, N, K;
    int wave, prow, pslot;
    bool last_w;            // this wavefront owns a W piece in the last (possibly partial) deal
    unsigned a_off[A_PER], w_off[W_PER];
    __device__ __forceinline__ void aim(int m0, int n0) {
#pragma unroll
        for (int i = 0; i < A_PER; i++) {
            const int r = 16 * (wave + NWAVES * i) + prow;                 // tile row of A
            a_off[i] = (unsigned)min(m0 + r, M - 1) * (unsigned)K + 4u * (pslot ^ ((r >> 2) & 3));
        }
#pragma unroll
        for (int i = 0; i < W_PER; i++) {
            const int r = 16 * (wave + NWAVES * i) + prow;                 // tile row of W (rows >= BN are never issued)
            w_off[i] = (unsigned)min(n0 + r, N - 1) * (unsigned)K + 4u * (pslot ^ ((r >> 2) & 3));
        }
    }
    __device__ __forceinline__ void issue(int kc, float* dst) const {
        const float* ap = A + kc * BK;
        const float* wp = W + kc * BK;
#pragma unroll
        for (int i = 0; i < A_PER; i++)
            __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(ap + a_off[i]), SCP_LDS_PTR(dst + (wave + NWAVES * i) * 256), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < W_PER; i++)
            if (i + 1 < W_PER || last_w)
                __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(wp + w_off[i]), SCP_LDS_PTR(dst + W_BASE + (wave + NWAVES * i) * 256), 16, 0, 0);
    }
    // "the chunk issued before the youngest one has landed": all but this wavefront's last stage issue are complete
    __device__ __forceinline__ void wait_older() const {
        constexpr int FULL = A_PER + W_PER;
        if (last_w) {
            if (FULL == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (FULL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (FULL == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            if (FULL == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (FULL == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (FULL == 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
    }
};

// One output tile: main loop over K, prologue of the next tile, epilogue.  TM x TN = MFMA tiles of this wavefront: 2 x 3 in a
// full panel; 1 x 1 in the remainder strip (the strip's 32 rows x 32 columns per wavefront for the first STRIP_WAVES
// wavefronts; the others only move their DMA pieces and keep the barriers).  The two cases are separate instantiations so
// that the accumulators never meet at a control-flow join (a join of two 96-register tuples made the allocator spill).
template <int EPI, int TM, int TN>
__device__ __forceinline__ void run_tile(const GemmArgs& g, Feeder& f, const Tile cur, bool has_next, const Tile nxt, float* lds0,
                                         float* lds1, float* lds2, int row_base, int col_base, bool active) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int nk = g.K / BK;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // lane's read offsets (bytes) inside a stage: rows row_base + 32*i + l31 of A, col_base + 32*j + l31 of W; chunk 2*half
    // (+ c: chunk c = 1 of a lane is its chunk 0 address with bit 4 flipped, the XOR swizzle)
    unsigned a_rd[TM], w_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int ra = row_base + 32 * i + l31;
        a_rd[i] = 4u * (ra * BK + 4 * ((2 * half) ^ ((ra >> 2) & 3)));
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int rw = col_base + 32 * j + l31;
        w_rd[j] = 4u * (W_BASE + rw * BK + 4 * ((2 * half) ^ ((rw >> 2) & 3)));
    }
    // LDS reads of the operand fragments are written as ds_read_b128 instructions by hand: for a compiler-visible LDS load
    // the wait-count pass assumes it may alias the LDS-DMA in flight and puts s_waitcnt vmcnt(0) in front of it, which
    // serialises the two-chunk prefetch.  The "+v" no-ops after the explicit lgkmcnt(0) make every fragment depend on it, so
    // that no MFMA is scheduled above the wait.
    auto compute_stage = [&](const float* stage) {
        if (!active) return;
        const unsigned base = (unsigned)(size_t)SCP_LDS_PTR(stage);
        f32x4 av[TM][2], wv[TN][2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int i = 0; i < TM; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(av[i][c]) : "v"((base + a_rd[i]) ^ (16u * c)));
#pragma unroll
            for (int j = 0; j < TN; j++) asm volatile("ds_read_b128 %0, %1" : "=v"(wv[j][c]) : "v"((base + w_rd[j]) ^ (16u * c)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int i = 0; i < TM; i++) asm volatile("" : "+v"(av[i][c]));
#pragma unroll
            for (int j = 0; j < TN; j++) asm volatile("" : "+v"(wv[j][c]));
        }
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].x, wv[j][c].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].y, wv[j][c].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].z, wv[j][c].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].w, wv[j][c].w, acc[i][j], 0, 0, 0);
                }
    };
    // three-stage ring, prefetch distance two chunks: while chunk kc is multiplied, kc+1 has been in flight for a whole
    // chunk time and kc+2 is issued.  Chunks 0 and 1 were issued by the prologue (of the kernel, or of the previous tile's
    // epilogue phase).
    auto step = [&](int kc, const float* stage, float* next) {
        if (kc + 1 < nk) f.wait_older();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // chunk kc visible to all; every wavefront is done with chunk kc-1, whose stage is refilled now.  A bare s_barrier:
        // __syncthreads() carries a workgroup release fence, for which the compiler waits for ALL outstanding LDS-DMA
        // (vmcnt(0)) -- that would cut the prefetch distance from two chunks to one.  The LDS reads of chunk kc-1 were
        // consumed by its MFMAs (lgkmcnt(0)), so nothing of this wavefront is in flight on the stage that is refilled.
        __builtin_amdgcn_s_barrier();
        if (kc + 2 < nk) f.issue(kc + 2, next);
        compute_stage(stage);
    };
    for (int kc = 0; kc < nk; kc += 3) {
        step(kc, lds0, lds2);
        if (kc + 1 < nk) step(kc + 1, lds1, lds0);
        if (kc + 2 < nk) step(kc + 2, lds2, lds1);
    }

    // ---- next tile: start its first two chunks before this tile's epilogue (every wavefront is done with the LDS ring; the
    // ring position restarts at stage 0 for every tile)
    __builtin_amdgcn_s_barrier();
    if (has_next) {
        f.aim(nxt.m0, nxt.n0);
        f.issue(0, lds0);
        if (nk > 1) f.issue(1, lds1);
    }

    // ---- epilogue.  MFMA layout: A operand rows -> accumulator rows acc_row(reg, half), B operand rows (W rows = output
    // columns) -> lane & 31: lane holds C[m][n = n_base + l31] for 16 rows m -> 32 consecutive floats per row per half-wave.
    // All loads of a 32 x 32 tile are issued before its stores (resid may alias C element for element; every element is read
    // and written by the same lane only).
    if (!active) return;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mb = cur.m0 + row_base + 32 * i;
        float mean[16], rstd[16];
        if (EPI == SCP_GEMM_LN || EPI == SCP_GEMM_LN_GELU) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = min(mb + acc_row(r, half), g.M - 1);
                const float2 st = *reinterpret_cast<const float2*>(g.rowstat + 2 * (size_t)m);
                mean[r] = st.x;
                rstd[r] = st.y;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = cur.n0 + col_base + 32 * j + l31;
            const bool n_ok = n < g.N;
            const int nc = min(n, g.N - 1);
            const float v0 = g.vec0[nc];
            float v1 = 0.f;
            if (EPI == SCP_GEMM_LN || EPI == SCP_GEMM_LN_GELU) v1 = g.vec1[nc];
            float res[16];
            if (EPI == SCP_GEMM_BIAS_RESIDUAL) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = min(mb + acc_row(r, half), g.M - 1);
                    res[r] = g.resid[(size_t)m * g.N + nc];
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
                if (m < g.M && n_ok) g.C[(size_t)m * g.N + n] = x;
            }
            __builtin_amdgcn_sched_barrier(0);      // one 32 x 32 tile at a time: keeps the next tile's loads out of this one's registers
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(THREADS, SCP_GEMM_WG_PER_CU) void vit_gemm_kernel(const GemmArgs g) {
    // one LDS object per stage (the compiler's wait-count insertion tracks LDS-DMA writes per object)
    __shared__ __attribute__((aligned(16))) float lds0[STAGE_FLOATS];
    __shared__ __attribute__((aligned(16))) float lds1[STAGE_FLOATS];
    __shared__ __attribute__((aligned(16))) float lds2[STAGE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    // Tile order.  (1) Full 256-row panels in an XCD-aware order: workgroup b runs on XCD b % 8, consecutive logical ids of
    // one XCD (adjacent in time) walk the N-blocks of one A panel, and every XCD gets the same number of full tiles.
    // (2) The N-blocks of a SHORT last panel (M = B * 1025 tokens leaves 32 rows) come last, in strip mode.  The kernel is
    // PERSISTENT: the grid is one workgroup per CU and every workgroup walks tiles t = blockIdx.x, + gridDim.x, ...; the
    // LDS-DMA prologue of its next tile is issued before the epilogue of the current one.
    const int full_slots = g.per_xcd * 8, total = full_slots + g.rem_blocks;
    auto tile_of = [&](int t) {
        Tile x;
        x.rem = t >= full_slots;
        int bm, bn;
        if (x.rem) {
            bn = t - full_slots;
            bm = g.full_panels;
            x.ok = true;
        } else {
            const int lid = (t & 7) * g.per_xcd + (t >> 3);
            x.ok = lid < g.full_panels * g.nblk_n;
            bm = lid / g.nblk_n;
            bn = lid - bm * g.nblk_n;
        }
        x.m0 = bm * BM;
        x.n0 = bn * BN;
        return x;
    };
    Feeder f;
    f.A = g.A; f.W = g.W; f.M = g.M; f.N = g.N; f.K = g.K;
    f.wave = wave; f.prow = lane >> 2; f.pslot = lane & 3;
    f.last_w = wave + NWAVES * (W_PER - 1) < W_PIECES;

    int t = blockIdx.x;
    Tile cur = tile_of(t);
    while (t < total && !cur.ok) { t += gridDim.x; if (t < total) cur = tile_of(t); }
    if (t >= total) return;
    f.aim(cur.m0, cur.n0);
    f.issue(0, lds0);
    if (g.K / BK > 1) f.issue(1, lds1);

    while (true) {
        int tn = t + gridDim.x;
        Tile nxt = cur;
        bool has_next = false;
        while (tn < total) {
            nxt = tile_of(tn);
            if (nxt.ok) { has_next = true; break; }
            tn += gridDim.x;
        }
        if (!cur.rem) {
            run_tile<EPI, WM, WN>(g, f, cur, has_next, nxt, lds0, lds1, lds2, 32 * WM * wm, 32 * WN * wn, true);
        } else {
            const bool active = wave < STRIP_WAVES;
            run_tile<EPI, 1, STRIP_TN>(g, f, cur, has_next, nxt, lds0, lds1, lds2, 0, active ? 32 * STRIP_TN * wave : 0, active);
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

// resident workgroup slots of the device: one workgroup per CU (84 KiB of LDS), a multiple of 8 so that tile t and tile
// t + grid land on the same XCD
int resident_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0)
            cus = 256;
        slots = max(8, (SCP_GEMM_WG_PER_CU * cus) & ~7);
    }
    return slots;
}

template <int EPI>
void launch(const GemmArgs& g, hipStream_t st) {
    const int total = g.per_xcd * 8 + g.rem_blocks;
    hipLaunchKernelGGL(vit_gemm_kernel<EPI>, dim3(min(total, resident_slots())), dim3(THREADS), 0, st, g);
}

}  // namespace

extern "C" int scp_vit_linear(const float* A, const float* W, const float* vec0, const float* vec1, const float* rowstat,
                              const float* resid, float* C, int M, int N, int K, int epilogue, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return scp::fail(hipErrorInvalidValue, "vit_linear: empty problem");
    if (K % (2 * BK) != 0) return scp::fail(hipErrorInvalidValue, "vit_linear: K must be a multiple of 32");
    if ((size_t)M * (size_t)K >= (1ull << 32) || (size_t)N * (size_t)K >= (1ull << 32))
        return scp::fail(hipErrorInvalidValue, "vit_linear: operand larger than 2^32 elements");
    const bool ln = epilogue == SCP_GEMM_LN || epilogue == SCP_GEMM_LN_GELU;
    if (!vec0 || (ln && (!vec1 || !rowstat)) || (epilogue == SCP_GEMM_BIAS_RESIDUAL && !resid))
        return scp::fail(hipErrorInvalidValue, "vit_linear: missing epilogue operand");
    GemmArgs g{};
    g.A = A; g.W = W; g.vec0 = vec0; g.vec1 = vec1; g.rowstat = rowstat; g.resid = resid; g.C = C;
    g.M = M; g.N = N; g.K = K;
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

extern "C" int scp_row_mean_rstd(const float* x, float* stats, int rows, int C, float eps, void* stream) {
    if (rows <= 0) return 0;
    if (C <= 0 || C > 1536 || (C & 3)) return scp::fail(hipErrorInvalidValue, "row_mean_rstd: C must be a multiple of 4 in 4..1536");
    hipLaunchKernelGGL(row_stats_kernel, dim3((rows + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), x, stats, rows,
                       C, eps);
    return scp::check_launch("row_mean_rstd");
}
