// self-corr-pose_amd/csrc/corr_pp.hip -- pixel <-> pixel correspondence of the rotation-cycle loss (a10) without the score tensor.
//
// Replaces model/module/correspondence.py:105-110: pc = src_feat^T @ tgt_feat [N,P,Q], masked to -1e5 where the source or the
// target pixel is background, softmax over the SOURCE pixels (dim P), cycle_match = grid @ softmax -> [N,2,Q]; and its backward.
// The reference (and the earlier build: library GEMM + csrc/corr.hip reductions + two library GEMMs backward) materialises the
// [N,P,Q] scores (134 MB at N = 32, P = Q = 1024) and goes through them four times.  Here a 32 x 32 score tile is a K = 64 MFMA
// accumulator that never leaves the registers:
//
//   forward   lane = target pixel q, registers = source pixels p: the softmax over p is lane local and online (running max /
//             sum / sum * grid); the four wavefronts of a workgroup take every fourth source tile and merge their states in LDS.
//   backward  d S[p][q] = tau P[p][q] (g_q . grid_p - g_q . out_q), zero where masked, rebuilt per tile from the saved
//             (max, sum) and used AS IS as the B operand of a second MFMA (the accumulator layout of v_mfma_f32_32x32x2_f32 is
//             its B layout): with lane = q the contraction runs over p -> g_tgt[c][q]; the same kernel with the roles of the
//             two feature maps exchanged (lane = p, registers = q) contracts over q -> g_src[c][p].  No atomics, fixed orders.
//
// Restrictions (else scp_amd/ops.py keeps the unfused path): C = 64 channels, P and Q multiples of 32.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef PP_FWD_WAVES
#define PP_FWD_WAVES 2      // wavefronts per SIMD the register allocation aims at (tools/build_fvm_variants.sh times the alternatives)
#endif
#ifndef PP_BWD_WAVES
#define PP_BWD_WAVES 2
#endif
constexpr int C = 64;
constexpr int IROW = 33;          // LDS row stride of a [64 channels][32 pixels] tile: conflict-free along channels and along pixels
constexpr float MASKED_SCORE = -1e5f;   // correspondence.py:106

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float other_half(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

struct PpArgs {
    const float* src;        // [N,64,P]
    const float* tgt;        // [N,64,Q]
    const float* src_mask;   // [N,P] or null
    const float* tgt_mask;   // [N,Q] or null
    const float* grid;       // [2,P] or [N,2,P]
    int grid_batched;
    float tau;
    int N, P, Q;
    float* out;              // [N,2,Q]   forward output / backward input
    float* stats;            // [N,2,Q]   (max of tau * s over p, sum of exp)
    const float* g_out;      // [N,2,Q]
    float* g_src;            // [N,64,P]
    float* g_tgt;            // [N,64,Q]
};

// the lane's pixel as the B operand of the score tile: breg[4t + r] = feat[c = 8t + 4 half + r][pixel]
__device__ __forceinline__ void load_lane_operand(const float* feat, int n, int npix, int pixel, int half, float* breg) {
    const float* fp = feat + (size_t)n * C * npix + pixel;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) breg[4 * t + r] = fp[(size_t)(8 * t + 4 * half + r) * npix];
}

// the tile's A operand straight from the registers a lane loaded (areg[4t + r] = feat[c = 8t + 4 half + r][p0 + l31], the same
// layout as the lane operand): acc[r] = sum_c feat_reg[c][pixel acc_row(r, half)] * lane_feat[c][lane's pixel]
__device__ __forceinline__ f32x16 score_tile(const float* areg, const float* breg) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[i], breg[i], acc, 0, 0, 0);
    return acc;
}

// the same registers as the wavefront's LDS tile[c][p] (read back column-wise by the second product)
__device__ __forceinline__ void store_tile(const float* areg, float* tile, int l31, int half) {
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) tile[(8 * t + 4 * half + r) * IROW + l31] = areg[4 * t + r];
}

// ------------------------------------------------------------------------------------------------------------------
// forward: grid (Q / 32, N), 4 wavefronts; lane = target pixel
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, PP_FWD_WAVES) void pp_forward_kernel(const PpArgs a) {
    __shared__ __attribute__((aligned(16))) float pt[4][32 * 4];        // per source pixel of the tile: grid x, grid y, foreground
    __shared__ __attribute__((aligned(16))) float red[4][32 * 4];

    const int n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int q = 32 * blockIdx.x + l31;
    const bool colok = a.tgt_mask == nullptr || a.tgt_mask[(size_t)n * a.Q + q] > 0.f;
    float breg[32];
    load_lane_operand(a.tgt, n, a.Q, q, half, breg);
    const float* gx = a.grid + (a.grid_batched ? (size_t)n * 2 * a.P : 0);
    const float* gy = gx + a.P;

    float m_run = -INFINITY, s_run = 0.f, ax = 0.f, ay = 0.f;
    float* my_pt = pt[wave];
    const int ntile = a.P / 32;
    float areg[32];
    if (wave < ntile) load_lane_operand(a.src, n, a.P, 32 * wave + l31, half, areg);
    for (int tp = wave; tp < ntile; tp += 4) {
        const int p0 = 32 * tp;
        if (lane < 32) {
            const int p = p0 + lane;
            const bool rowok = a.src_mask == nullptr || a.src_mask[(size_t)n * a.P + p] > 0.f;
            *reinterpret_cast<float4*>(my_pt + 4 * lane) = make_float4(gx[p], gy[p], rowok ? 1.f : 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x16 acc = score_tile(areg, breg);
        if (tp + 4 < ntile) load_lane_operand(a.src, n, a.P, p0 + 128 + l31, half, areg);     // next tile, under this tile's softmax
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float ok = my_pt[4 * acc_row(r, half) + 2];
            const float x = a.tau * ((colok && ok > 0.f) ? acc[r] : MASKED_SCORE);
            acc[r] = x;
            tmax = fmaxf(tmax, x);
        }
        if (tmax > m_run) {
            const float sc = expf(m_run - tmax);          // exp(-inf) = 0 the first time
            s_run *= sc; ax *= sc; ay *= sc;
            m_run = tmax;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float2 g = *reinterpret_cast<const float2*>(my_pt + 4 * acc_row(r, half));
            const float e = expf(acc[r] - m_run);
            s_run += e; ax += e * g.x; ay += e * g.y;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the two halves hold disjoint source pixels of the same target pixel; then the four wavefronts (fixed order)
    {
        const float m_o = other_half(m_run), s_o = other_half(s_run), ax_o = other_half(ax), ay_o = other_half(ay);
        const float M = fmaxf(m_run, m_o);
        const float s1 = m_run == -INFINITY ? 0.f : expf(m_run - M), s2 = m_o == -INFINITY ? 0.f : expf(m_o - M);
        if (half == 0) *reinterpret_cast<float4*>(red[wave] + 4 * l31) = make_float4(M, s_run * s1 + s_o * s2, ax * s1 + ax_o * s2, ay * s1 + ay_o * s2);
    }
    __syncthreads();
    if (tid < 32) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; w++) M = fmaxf(M, red[w][4 * tid]);
        float s = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float4 c4 = *reinterpret_cast<const float4*>(red[w] + 4 * tid);
            const float sc = c4.x == -INFINITY ? 0.f : expf(c4.x - M);
            s += c4.y * sc; sx += c4.z * sc; sy += c4.w * sc;
        }
        const size_t o = (size_t)n * 2 * a.Q + 32 * blockIdx.x + tid;
        a.out[o] = sx / s;
        a.out[o + a.Q] = sy / s;
        a.stats[o] = M;
        a.stats[o + a.Q] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward: grid (lane-side pixels / 32, N), 4 wavefronts.  QL: lane = target pixel q, registers = source pixels -> g_tgt;
// !QL: lane = source pixel p, registers = target pixels -> g_src.
// ------------------------------------------------------------------------------------------------------------------
template <bool QL>
__global__ __launch_bounds__(256, PP_BWD_WAVES) void pp_backward_kernel(const PpArgs a) {
    static_assert(C * IROW >= 32 * 65, "the final cross-wavefront sum reuses the feature tiles");
    __shared__ __attribute__((aligned(16))) float it[4][C * IROW];      // per wavefront: register-side feature tile; at the end its partial sums
    __shared__ __attribute__((aligned(16))) float pt[4][32 * 8];        // per register-side pixel: its scalars of the d S formula

    const int n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int n_lane = QL ? a.Q : a.P, n_reg = QL ? a.P : a.Q;
    const float* lane_feat = QL ? a.tgt : a.src;
    const float* reg_feat = QL ? a.src : a.tgt;
    const int li = 32 * blockIdx.x + l31;
    const float* gx = a.grid + (a.grid_batched ? (size_t)n * 2 * a.P : 0);
    const float* gy = gx + a.P;
    const size_t qbase = (size_t)n * 2 * a.Q;

    // q-side scalars of pixel q: softmax max, 1 / sum, the upstream gradient (2) and its dot with the output; p-side: grid (2)
    auto q_scalars = [&](int q, float4& s0, float2& s1) {
        const float g0 = a.g_out[qbase + q], g1 = a.g_out[qbase + a.Q + q];
        const bool ok = a.tgt_mask == nullptr || a.tgt_mask[(size_t)n * a.Q + q] > 0.f;
        s0 = make_float4(a.stats[qbase + q], 1.f / a.stats[qbase + a.Q + q], g0, g1);
        s1 = make_float2(g0 * a.out[qbase + q] + g1 * a.out[qbase + a.Q + q], ok ? 1.f : 0.f);
    };
    auto p_scalars = [&](int p, float4& s0) {
        const bool ok = a.src_mask == nullptr || a.src_mask[(size_t)n * a.P + p] > 0.f;
        s0 = make_float4(gx[p], gy[p], ok ? 1.f : 0.f, 0.f);
    };

    float breg[32];
    load_lane_operand(lane_feat, n, n_lane, li, half, breg);
    float4 l0;
    float2 l1 = make_float2(0.f, 0.f);
    if (QL) q_scalars(li, l0, l1);
    else p_scalars(li, l0);
    const bool lane_ok = QL ? l1.y > 0.f : l0.z > 0.f;

    f32x16 g_lo, g_hi;
#pragma unroll
    for (int r = 0; r < 16; r++) { g_lo[r] = 0.f; g_hi[r] = 0.f; }
    float* my_it = it[wave];
    float* my_pt = pt[wave];
    const int ntile = n_reg / 32;
    float areg[32];
    if (wave < ntile) load_lane_operand(reg_feat, n, n_reg, 32 * wave + l31, half, areg);
    for (int tr = wave; tr < ntile; tr += 4) {
        const int r0 = 32 * tr;
        store_tile(areg, my_it, l31, half);
        if (lane < 32) {
            float4 s0;
            float2 s1 = make_float2(0.f, 0.f);
            if (QL) p_scalars(r0 + lane, s0);
            else q_scalars(r0 + lane, s0, s1);
            *reinterpret_cast<float4*>(my_pt + 8 * lane) = s0;
            *reinterpret_cast<float2*>(my_pt + 8 * lane + 4) = s1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x16 acc = score_tile(areg, breg);                   // S[p][q] at (register pixel, lane pixel)
        if (tr + 4 < ntile) load_lane_operand(reg_feat, n, n_reg, r0 + 128 + l31, half, areg);   // next tile, under d S and the second product
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float* o = my_pt + 8 * acc_row(r, half);
            const float4 o0 = *reinterpret_cast<const float4*>(o);
            float ds = 0.f;
            if (QL) {
                if (lane_ok && o0.z > 0.f)
                    ds = a.tau * (expf(a.tau * acc[r] - l0.x) * l0.y) * (l0.z * o0.x + l0.w * o0.y - l1.x);
            } else {
                const float2 o1 = *reinterpret_cast<const float2*>(o + 4);
                if (lane_ok && o1.y > 0.f)
                    ds = a.tau * (expf(a.tau * acc[r] - o0.x) * o0.y) * (o0.z * l0.x + o0.w * l0.y - o1.x);
            }
            acc[r] = ds;
        }
        // g[c][lane pixel] += sum over the tile's register pixels of reg_feat[c][pixel] * d S: k-step r pairs the pixels
        // acc_row(r, 0) and acc_row(r, 1); the A operand is reg_feat[c = l31 (+32)][acc_row(r, half)]
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pr = acc_row(r, half);
            g_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(my_it[l31 * IROW + pr], acc[r], g_lo, 0, 0, 0);
            g_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(my_it[(l31 + 32) * IROW + pr], acc[r], g_hi, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- sum the four wavefronts (fixed order); the lane holds [c = acc_row(r, half) (+32)][pixel l31].  red[wave] is the
    //      wavefront's own tile, which only it reads: no barrier needed before overwriting it
    float (*red)[C * IROW] = it;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        red[wave][l31 * 65 + acc_row(r, half)] = g_lo[r];
        red[wave][l31 * 65 + acc_row(r, half) + 32] = g_hi[r];
    }
    __syncthreads();
    float* g = (QL ? a.g_tgt : a.g_src) + (size_t)n * C * n_lane + 32 * blockIdx.x;
    for (int e = tid; e < 32 * C; e += 256) {
        const int c = e >> 5, pl = e & 31;
        g[(size_t)c * n_lane + pl] = (red[0][pl * 65 + c] + red[1][pl * 65 + c]) + (red[2][pl * 65 + c] + red[3][pl * 65 + c]);
    }
}

int check_shape(int N, int Cf, int P, int Q) {
    if (N <= 0 || P <= 0 || Q <= 0) return scp::fail(hipErrorInvalidValue, "pixel_pixel_softargmax: empty problem");
    if (Cf != C || (P & 31) || (Q & 31))
        return scp::fail(hipErrorInvalidValue, "pixel_pixel_softargmax (fused): needs 64 channels and pixel counts that are multiples of 32");
    return 0;
}

}  // namespace

extern "C" int scp_pp_softargmax_forward(const float* src_feat, const float* tgt_feat, const float* src_mask, const float* tgt_mask,
                                         const float* grid, int grid_batched, float tau, int N, int Cf, int P, int Q, float* out,
                                         float* colstats, void* stream) {
    if (int e = check_shape(N, Cf, P, Q)) return e;
    PpArgs a{};
    a.src = src_feat; a.tgt = tgt_feat; a.src_mask = src_mask; a.tgt_mask = tgt_mask; a.grid = grid; a.grid_batched = grid_batched;
    a.tau = tau; a.N = N; a.P = P; a.Q = Q; a.out = out; a.stats = colstats;
    hipLaunchKernelGGL(pp_forward_kernel, dim3(Q / 32, N), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return scp::check_launch("pp_forward");
}

extern "C" int scp_pp_softargmax_backward(const float* src_feat, const float* tgt_feat, const float* src_mask, const float* tgt_mask,
                                          const float* grid, int grid_batched, float tau, int N, int Cf, int P, int Q,
                                          const float* out, const float* colstats, const float* g_out, float* g_src_feat,
                                          float* g_tgt_feat, void* stream) {
    if (int e = check_shape(N, Cf, P, Q)) return e;
    PpArgs a{};
    a.src = src_feat; a.tgt = tgt_feat; a.src_mask = src_mask; a.tgt_mask = tgt_mask; a.grid = grid; a.grid_batched = grid_batched;
    a.tau = tau; a.N = N; a.P = P; a.Q = Q; a.out = const_cast<float*>(out); a.stats = const_cast<float*>(colstats); a.g_out = g_out;
    a.g_src = g_src_feat; a.g_tgt = g_tgt_feat;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_tgt_feat) {
        hipLaunchKernelGGL(pp_backward_kernel<true>, dim3(Q / 32, N), dim3(256), 0, st, a);
        if (int e = scp::check_launch("pp_backward_tgt")) return e;
    }
    if (g_src_feat) {
        hipLaunchKernelGGL(pp_backward_kernel<false>, dim3(P / 32, N), dim3(256), 0, st, a);
        if (int e = scp::check_launch("pp_backward_src")) return e;
    }
    return 0;
}
