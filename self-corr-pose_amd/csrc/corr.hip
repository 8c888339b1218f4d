// self-corr-pose_amd/csrc/corr.hip -- dense correspondence operators: masked softmax / soft-argmax
// over an all-pairs score tensor S[N,P,Q] (Q contiguous), forward and backward, for gfx950.
//
// What they compute (reference, model/module/):
//   correspondence.py:44-53   pc masked to -1e5; softmax over pixels (dim P) -> imatch = grid @ P_mesh;
//                             softmax over vertices (dim Q) -> match = P_img @ verts
//   correspondence.py:105-110 masked pixel x pixel scores, softmax over source pixels, grid @ P
//   pretrained_corr.py:123-137 the same two softmaxes on the 2x2-pooled scores
// The reference materialises every softmax (and a [B,P,V,3] temporary) and runs ~10 full passes over
// the 0.34-1.3 GB score tensor per direction.  These are HBM-bound reductions, so here:
//   * scores are read ONCE per reduction with 64 lanes on 64 consecutive columns (256 B per
//     wavefront instruction), online-softmax state in registers, no probability tensor exists;
//   * the column (over P) reduction is split into row chunks so that >= 5k workgroups are in flight
//     (a launch needs >> 256 workgroups to fill the chip), partial (max, sum, sum*gx, sum*gy) states are
//     merged by a tiny second kernel;
//   * the row (over Q) reduction keeps the row in registers (one wavefront per row, DPP/shuffle
//     reductions);
//   * backward recomputes both probabilities from the saved (max, sum) statistics in one fused
//     element-wise pass: d S = g_in + tau_c P_c (g_c . grid_p - g_c . out_c) + tau_r P_r (g_r . w_q - g_r . out_r).
// No MFMA here by design: arithmetic intensity is < 1 flop/byte.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr float MASKED_SCORE = -1e5f;  // correspondence.py:44, pretrained_corr.py:86
constexpr int COLS_ROWS_PER_BLOCK = 256;

struct OnlineState {
    float m, s, ax, ay;
};

__device__ __forceinline__ void online_push(OnlineState& st, float x, float gx, float gy) {
    if (x > st.m) {
        const float sc = expf(st.m - x);  // exp(-inf) = 0 on the first element
        st.s = st.s * sc + 1.f;
        st.ax = st.ax * sc + gx;
        st.ay = st.ay * sc + gy;
        st.m = x;
    } else {
        const float e = expf(x - st.m);
        st.s += e;
        st.ax += e * gx;
        st.ay += e * gy;
    }
}

__device__ __forceinline__ void online_merge(OnlineState& a, const OnlineState& b) {
    const float m = fmaxf(a.m, b.m);
    const float sa = a.m == -INFINITY ? 0.f : expf(a.m - m);
    const float sb = b.m == -INFINITY ? 0.f : expf(b.m - m);
    a.s = a.s * sa + b.s * sb;
    a.ax = a.ax * sa + b.ax * sb;
    a.ay = a.ay * sa + b.ay * sb;
    a.m = m;
}

// ---- column soft-argmax, pass 1: partial states per (n, chunk, q) --------------------------------
// grid: blockIdx.x = q tile (64 columns), blockIdx.y = row chunk, blockIdx.z = n.  4 wavefronts,
// wavefront w takes rows chunk*256 + w, w+4, ...
__global__ __launch_bounds__(256) void cols_partial_kernel(
    const float* __restrict__ S, float* __restrict__ S_out, const float* __restrict__ rowmask,
    const float* __restrict__ colmask, const float* __restrict__ grid, int grid_batched, float tau, int P,
    int Q, int nchunk, float* __restrict__ partial) {
    __shared__ OnlineState sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const int chunk = blockIdx.y, n = blockIdx.z;
    const bool qok = q < Q;
    const bool colok = qok && (colmask == nullptr || colmask[(size_t)n * Q + q] > 0.f);
    const float* gx = grid + (grid_batched ? (size_t)n * 2 * P : 0);
    const float* gy = gx + P;
    const float* rm = rowmask ? rowmask + (size_t)n * P : nullptr;
    const size_t base = (size_t)n * P * Q;
    OnlineState st{-INFINITY, 0.f, 0.f, 0.f};
    const int r0 = chunk * COLS_ROWS_PER_BLOCK;
    const int r1 = min(r0 + COLS_ROWS_PER_BLOCK, P);
    // eight rows per pass: their loads are issued together (the online update is a dependent chain with a branch -- issued row by
    // row, every row paid the full memory latency), then pushed in row order
    for (int rb = r0 + wave; rb < r1; rb += 32) {
        float v[8], g0[8], g1[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int r = rb + 4 * u;
            ok[u] = r < r1;
            const int rc = ok[u] ? r : r0;
            const bool rowok = rm == nullptr || rm[rc] > 0.f;   // wavefront-uniform
            v[u] = qok ? S[base + (size_t)rc * Q + q] : 0.f;
            if (!(rowok && colok)) v[u] = MASKED_SCORE;
            g0[u] = gx[rc]; g1[u] = gy[rc];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (ok[u] && qok) {
                if (S_out) S_out[base + (size_t)(rb + 4 * u) * Q + q] = v[u];
                online_push(st, tau * v[u], g0[u], g1[u]);
            }
        }
    }
    sh[wave][lane] = st;
    __syncthreads();
    if (wave == 0 && qok) {
        online_merge(st, sh[1][lane]);
        online_merge(st, sh[2][lane]);
        online_merge(st, sh[3][lane]);
        float* o = partial + (((size_t)n * nchunk + chunk) * 4) * Q + q;
        o[0] = st.m; o[Q] = st.s; o[2 * Q] = st.ax; o[3 * Q] = st.ay;
    }
}

// ---- column soft-argmax, pass 2: merge the chunks ------------------------------------------------
__global__ __launch_bounds__(256) void cols_combine_kernel(const float* __restrict__ partial, int Q,
                                                           int nchunk, float* __restrict__ out,
                                                           float* __restrict__ stats) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (q >= Q) return;
    OnlineState st{-INFINITY, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunk; c++) {
        const float* o = partial + (((size_t)n * nchunk + c) * 4) * Q + q;
        const OnlineState b{o[0], o[Q], o[2 * Q], o[3 * Q]};
        online_merge(st, b);
    }
    out[((size_t)n * 2 + 0) * Q + q] = st.ax / st.s;
    out[((size_t)n * 2 + 1) * Q + q] = st.ay / st.s;
    stats[((size_t)n * 2 + 0) * Q + q] = st.m;   // max of tau * score
    stats[((size_t)n * 2 + 1) * Q + q] = st.s;
}

// ---- row softmax with a weighted sum: one wavefront per row, the row lives in registers ----------
template <int W>
__device__ __forceinline__ void wave_sum_vec(float* v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
#pragma unroll
        for (int k = 0; k < W; k++) v[k] += __shfl_xor(v[k], m);
}

constexpr int ROW_CACHE = 16;  // 64 lanes x 16 = rows up to 1024 columns stay in registers

template <int W>
__global__ __launch_bounds__(256) void rows_weighted_kernel(const float* __restrict__ S,
                                                            const float* __restrict__ weights, float tau,
                                                            int P, int Q, long rows,
                                                            float* __restrict__ out, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n = (int)(row / P);
    const float* s = S + (size_t)row * Q;
    const float* w = weights + (size_t)n * Q * W;
    float x[ROW_CACHE];
    float mx = -INFINITY;
    const bool cached = Q <= 64 * ROW_CACHE;
    if (cached) {
#pragma unroll
        for (int i = 0; i < ROW_CACHE; i++) {
            const int q = lane + 64 * i;
            x[i] = q < Q ? tau * s[q] : -INFINITY;
            mx = fmaxf(mx, x[i]);
        }
    } else {
        for (int q = lane; q < Q; q += 64) mx = fmaxf(mx, tau * s[q]);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    float acc[W + 1];
#pragma unroll
    for (int k = 0; k <= W; k++) acc[k] = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < ROW_CACHE; i++) {
            const int q = lane + 64 * i;
            if (q < Q) {
                const float e = expf(x[i] - mx);
                acc[W] += e;
#pragma unroll
                for (int k = 0; k < W; k++) acc[k] += e * w[(size_t)q * W + k];
            }
        }
    } else {
        for (int q = lane; q < Q; q += 64) {
            const float e = expf(tau * s[q] - mx);
            acc[W] += e;
#pragma unroll
            for (int k = 0; k < W; k++) acc[k] += e * w[(size_t)q * W + k];
        }
    }
    wave_sum_vec<W + 1>(acc);
    if (lane < W) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < W; k++) r = lane == k ? acc[k] : r;
        out[(size_t)row * W + lane] = r / acc[W];
    } else if (lane == W) {
        stats[(size_t)row * 2 + 0] = mx;
        stats[(size_t)row * 2 + 1] = acc[W];
    }
}

// ---- fused backward of both reductions -----------------------------------------------------------
struct DualBwdArgs {
    const float* S;          // [N,P,Q] scores as the forward reductions saw them (already masked)
    const float* rowmask;    // [N,P] or null
    const float* colmask;    // [N,Q] or null
    const float* g_in;       // [N,P,Q] or null: gradient arriving directly on the scores
    float* g_out;            // [N,P,Q]
    // column (over P) soft-argmax
    const float* cstats;     // [N,2,Q] or null when the column part is absent
    const float* cout;       // [N,2,Q]
    const float* g_cout;     // [N,2,Q]
    const float* grid;       // [grid_n,2,P]
    int grid_batched;
    float tau_c;
    // row (over Q) weighted softmax
    const float* rstats;     // [N,P,2] or null
    const float* rout;       // [N,P,W]
    const float* g_rout;     // [N,P,W]
    const float* weights;    // [N,Q,W]
    float tau_r;
    int P, Q;
};

template <int W>
__global__ __launch_bounds__(256) void dual_backward_kernel(const DualBwdArgs a) {
    const long row = blockIdx.x;           // n*P + p
    const int n = (int)(row / a.P), p = (int)(row - (long)n * a.P);
    const bool rowok = a.rowmask == nullptr || a.rowmask[row] > 0.f;
    const float* s = a.S + (size_t)row * a.Q;
    float* go = a.g_out + (size_t)row * a.Q;
    const float* gi = a.g_in ? a.g_in + (size_t)row * a.Q : nullptr;
    // row scalars (uniform over the block)
    float gx = 0.f, gy = 0.f;
    if (a.cstats) {
        const float* g = a.grid + (a.grid_batched ? (size_t)n * 2 * a.P : 0);
        gx = g[p]; gy = g[a.P + p];
    }
    float rmax = 0.f, rinv = 0.f, gr[W], rdot = 0.f;
#pragma unroll
    for (int k = 0; k < W; k++) gr[k] = 0.f;
    if (a.rstats) {
        rmax = a.rstats[(size_t)row * 2];
        rinv = 1.f / a.rstats[(size_t)row * 2 + 1];
#pragma unroll
        for (int k = 0; k < W; k++) {
            gr[k] = a.g_rout[(size_t)row * W + k];
            rdot += gr[k] * a.rout[(size_t)row * W + k];
        }
    }
    for (int q = threadIdx.x; q < a.Q; q += blockDim.x) {
        const bool ok = rowok && (a.colmask == nullptr || a.colmask[(size_t)n * a.Q + q] > 0.f);
        float d = 0.f;
        if (ok) {
            const float v = s[q];
            if (gi) d = gi[q];
            if (a.cstats) {
                const size_t c = (size_t)n * 2 * a.Q + q;
                const float pc = expf(a.tau_c * v - a.cstats[c]) / a.cstats[c + a.Q];
                const float gcx = a.g_cout[c], gcy = a.g_cout[c + a.Q];
                d += a.tau_c * pc * (gcx * gx + gcy * gy - (gcx * a.cout[c] + gcy * a.cout[c + a.Q]));
            }
            if (a.rstats) {
                const float pr = expf(a.tau_r * v - rmax) * rinv;
                const float* w = a.weights + ((size_t)n * a.Q + q) * W;
                float dot = 0.f;
#pragma unroll
                for (int k = 0; k < W; k++) dot += gr[k] * w[k];
                d += a.tau_r * pr * (dot - rdot);
            }
        }
        go[q] = d;
    }
}

// ---- backward of the column soft-argmax alone (no row part, no direct gradient): thread = column, COLS_BWD_ROWS rows per workgroup.
// The column's constants (max, 1 / sum, upstream gradient, its dot with the output) stay in registers over the rows -- in the
// general kernel above every row's workgroup re-reads the six per-column arrays -- and each row is one coalesced read and one
// coalesced write.  d S = tau P (g . grid_p - g . out_q), zero where masked.
constexpr int COLS_BWD_ROWS = 32;
__global__ __launch_bounds__(256) void cols_backward_kernel(const DualBwdArgs a) {
    const int q = blockIdx.x * 256 + threadIdx.x, n = blockIdx.z;
    const int p0 = blockIdx.y * COLS_BWD_ROWS, p1 = min(p0 + COLS_BWD_ROWS, a.P);
    if (q >= a.Q) return;
    const size_t c = (size_t)n * 2 * a.Q + q;
    const bool colok = a.colmask == nullptr || a.colmask[(size_t)n * a.Q + q] > 0.f;
    const float cmax = a.cstats[c], cinv = 1.f / a.cstats[c + a.Q];
    const float gcx = a.g_cout[c], gcy = a.g_cout[c + a.Q];
    const float cdot = gcx * a.cout[c] + gcy * a.cout[c + a.Q];
    const float* g = a.grid + (a.grid_batched ? (size_t)n * 2 * a.P : 0);
    const float* rm = a.rowmask ? a.rowmask + (size_t)n * a.P : nullptr;
    const size_t base = (size_t)n * a.P * a.Q + q;
#pragma unroll 8
    for (int p = p0; p < p1; p++) {
        const bool ok = colok && (rm == nullptr || rm[p] > 0.f);
        float d = 0.f;
        if (ok) {
            const float pc = expf(a.tau_c * a.S[base + (size_t)p * a.Q] - cmax) * cinv;
            d = a.tau_c * pc * (gcx * g[p] + gcy * g[a.P + p] - cdot);
        }
        a.g_out[base + (size_t)p * a.Q] = d;
    }
}

}  // namespace

extern "C" size_t scp_softargmax_cols_workspace(int N, int P, int Q) {
    const size_t nchunk = (size_t)(P + COLS_ROWS_PER_BLOCK - 1) / COLS_ROWS_PER_BLOCK;
    return (size_t)N * nchunk * 4 * Q * sizeof(float);
}

extern "C" int scp_softargmax_cols_forward(const float* scores, float* scores_masked_out,
                                           const float* rowmask, const float* colmask, const float* grid,
                                           int grid_batched, float tau, int N, int P, int Q, float* out,
                                           float* colstats, float* workspace, size_t workspace_bytes,
                                           void* stream) {
    if (N <= 0 || P <= 0 || Q <= 0) return scp::fail(hipErrorInvalidValue, "softargmax_cols: empty problem");
    if (workspace_bytes < scp_softargmax_cols_workspace(N, P, Q))
        return scp::fail(hipErrorInvalidValue, "softargmax_cols: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunk = (P + COLS_ROWS_PER_BLOCK - 1) / COLS_ROWS_PER_BLOCK;
    hipLaunchKernelGGL(cols_partial_kernel, dim3((Q + 63) / 64, nchunk, N), dim3(256), 0, st, scores,
                       scores_masked_out, rowmask, colmask, grid, grid_batched, tau, P, Q, nchunk, workspace);
    if (int e = scp::check_launch("cols_partial")) return e;
    hipLaunchKernelGGL(cols_combine_kernel, dim3((Q + 255) / 256, N), dim3(256), 0, st, workspace, Q, nchunk,
                       out, colstats);
    return scp::check_launch("cols_combine");
}

extern "C" int scp_softmax_rows_weighted_forward(const float* scores, const float* weights, int W, float tau,
                                                 int N, int P, int Q, float* out, float* rowstats,
                                                 void* stream) {
    if (N <= 0 || P <= 0 || Q <= 0) return scp::fail(hipErrorInvalidValue, "softmax_rows: empty problem");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long rows = (long)N * P;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (W == 3) hipLaunchKernelGGL(rows_weighted_kernel<3>, grid, block, 0, st, scores, weights, tau, P, Q, rows, out, rowstats);
    else if (W == 2) hipLaunchKernelGGL(rows_weighted_kernel<2>, grid, block, 0, st, scores, weights, tau, P, Q, rows, out, rowstats);
    else return scp::fail(hipErrorInvalidValue, "softmax_rows: W must be 2 or 3");
    return scp::check_launch("rows_weighted");
}

extern "C" int scp_dual_softmax_backward(const float* scores, const float* rowmask, const float* colmask,
                                         const float* g_scores_in, float* g_scores_out,
                                         const float* colstats, const float* col_out, const float* g_col_out,
                                         const float* grid, int grid_batched, float tau_c,
                                         const float* rowstats, const float* row_out, const float* g_row_out,
                                         const float* weights, int W, float tau_r, int N, int P, int Q,
                                         void* stream) {
    if (N <= 0 || P <= 0 || Q <= 0) return scp::fail(hipErrorInvalidValue, "dual_softmax_backward: empty problem");
    DualBwdArgs a{scores, rowmask, colmask, g_scores_in, g_scores_out, colstats, col_out, g_col_out, grid,
                  grid_batched, tau_c, rowstats, row_out, g_row_out, weights, tau_r, P, Q};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rowstats == nullptr && g_scores_in == nullptr && colstats != nullptr) {
        hipLaunchKernelGGL(cols_backward_kernel, dim3((Q + 255) / 256, (P + COLS_BWD_ROWS - 1) / COLS_BWD_ROWS, N), dim3(256), 0, st, a);
        return scp::check_launch("cols_backward");
    }
    const dim3 grid_dim((unsigned)((long)N * P)), block(256);
    if (rowstats == nullptr || W == 3) hipLaunchKernelGGL(dual_backward_kernel<3>, grid_dim, block, 0, st, a);
    else if (W == 2) hipLaunchKernelGGL(dual_backward_kernel<2>, grid_dim, block, 0, st, a);
    else return scp::fail(hipErrorInvalidValue, "dual_softmax_backward: W must be 2 or 3");
    return scp::check_launch("dual_backward");
}
