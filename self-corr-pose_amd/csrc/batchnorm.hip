// self-corr-pose_amd/csrc/batchnorm.hip -- BatchNorm2d (+ residual add) (+ ReLU) for the ResNet18 trunk of the
// image encoder, NHWC fp32, forward and backward.
//
// Replaces the `relu(bn(conv(x)))` / `relu(bn(conv(y)) + skip)` chains of torchvision's BasicBlock as used
// by model/module/network/image_encoder.py:119-139 (called twice per step: encoder.py:31 and
// correspondence.py:91).  Through PyTorch these are MIOpen's three-kernel spatial BN + an in-place ReLU
// + an add in forward and threshold_backward + two BN kernels in backward: 5 (8 with the residual) passes
// over the activation forward and 8 backward.  All of it is HBM-bound streaming, so the budget here is
// passes over the activation [R = N*H*W rows, C channels]:
//     forward : stats (1 read) + apply (1 read [+1 skip], 1 write)                       = 3 (4)
//     backward: reduce (dy, x [, y] reads [, g write]) + dx (dy|g, x reads, 1 write)     = 5 (7)
// The ReLU mask is recomputed from x in the plain case (same expression as the forward, bit-identical), so
// the output never has to be re-read; with a residual the saved output gives the mask and the masked
// gradient g is written once (it is also the gradient of the skip branch).
// Statistics: fp32 partial sums per (block, channel) over <= a few hundred rows, folded in fp64.
#include <hip/hip_runtime.h>

#include "bn_common.h"
#include "scp_common.h"
#include "scp_hip.h"

namespace {

using namespace scp_bn;

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ x, long R, int C, int tc_n,
                                                              int rows_per_block, float* __restrict__ psum,
                                                              float* __restrict__ psq, unsigned* __restrict__ ticket,
                                                              const FwdFinalize fin) {
    __shared__ float4 lds[2 * BN_THREADS];
    const int ri_n = BN_THREADS / tc_n;
    const int tc = threadIdx.x % tc_n, ri = threadIdx.x / tc_n;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, R);
    float4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    for (long r = r0 + ri; r < r1; r += ri_n) {
        const float4 v = ld4(x + r * C + 4 * tc);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
    }
    fold_rows(s, q, tc_n, ri_n, psum, psq, C, lds);
    if (!last_block_arrived(ticket)) return;
    if (threadIdx.x == 0 && fin.batches_tracked) *fin.batches_tracked += 1;
    double sa[4], sb[4];
    if (!fold_partials(psum, psq, (int)gridDim.x, C, tc_n, sa, sb, lds)) return;
#pragma unroll
    for (int i = 0; i < 4; i++) finalize_channel(fin, 4 * tc + i, true, sa[i], sb[i]);
}

// eval mode: no batch statistics, one thread per channel
__global__ void bn_eval_finalize_kernel(int C, const FwdFinalize fin) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) finalize_channel(fin, c, false, 0.0, 0.0);
}

template <typename T, bool SKIP, bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x,
                                                              const T* __restrict__ skip, long quads, int tc_n,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              T* __restrict__ y) {
    const long i = (long)blockIdx.x * BN_THREADS + threadIdx.x;
    if (i >= quads) return;
    const int tc = (int)(i % tc_n);
    const float4 sc = ld4(scale + 4 * tc), sh = ld4(shift + 4 * tc);
    const float4 v = ld4(x + 4 * i);
    float4 o = {bn_apply(v.x, sc.x, sh.x), bn_apply(v.y, sc.y, sh.y), bn_apply(v.z, sc.z, sh.z),
                bn_apply(v.w, sc.w, sh.w)};
    if (SKIP) {
        const float4 k = ld4(skip + 4 * i);
        o.x += k.x; o.y += k.y; o.z += k.z; o.w += k.w;
    }
    if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4(y + 4 * i, o);
}

// MODE 0: no activation; 1: ReLU, mask recomputed from x; 2: ReLU, mask from the saved output (residual), g written
template <typename T, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, long R, int C, int tc_n,
    int rows_per_block, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ g_out,
    float* __restrict__ pg, float* __restrict__ pgx, unsigned* __restrict__ ticket, int training, float* __restrict__ c1,
    float* __restrict__ c2, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float4 lds[2 * BN_THREADS];
    const int ri_n = BN_THREADS / tc_n;
    const int tc = threadIdx.x % tc_n, ri = threadIdx.x / tc_n;
    const float4 mu = ld4(mean + 4 * tc), is = ld4(invstd + 4 * tc);
    float4 sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
    if (MODE == 1) { sc = ld4(scale + 4 * tc); sh = ld4(shift + 4 * tc); }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, R);
    float4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    for (long r = r0 + ri; r < r1; r += ri_n) {
        const size_t o = r * C + 4 * tc;
        float4 g = ld4(dy + o);
        const float4 v = ld4(x + o);
        if (MODE == 1) {
            if (!(bn_apply(v.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
            if (!(bn_apply(v.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
            if (!(bn_apply(v.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
            if (!(bn_apply(v.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
        } else if (MODE == 2) {
            const float4 out = ld4(y + o);
            if (!(out.x > 0.f)) g.x = 0.f;
            if (!(out.y > 0.f)) g.y = 0.f;
            if (!(out.z > 0.f)) g.z = 0.f;
            if (!(out.w > 0.f)) g.w = 0.f;
            st4(g_out + o, g);
        }
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        q.x = fmaf(g.x, (v.x - mu.x) * is.x, q.x);
        q.y = fmaf(g.y, (v.y - mu.y) * is.y, q.y);
        q.z = fmaf(g.z, (v.z - mu.z) * is.z, q.z);
        q.w = fmaf(g.w, (v.w - mu.w) * is.w, q.w);
    }
    fold_rows(s, q, tc_n, ri_n, pg, pgx, C, lds);
    if (!last_block_arrived(ticket)) return;
    double sa[4], sb[4];
    if (!fold_partials(pg, pgx, (int)gridDim.x, C, tc_n, sa, sb, lds)) return;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = 4 * tc + i;
        if (dgamma) dgamma[c] = (float)sb[i];
        if (dbeta) dbeta[c] = (float)sa[i];
        // eval mode: the statistics are constants, dx = scale * g
        c1[c] = training ? (float)(sa[i] / (double)R) : 0.f;
        c2[c] = training ? (float)(sb[i] / (double)R) : 0.f;
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_dx_kernel(const T* __restrict__ dy,
                                                               const T* __restrict__ x, long quads, int tc_n,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ c1,
                                                               const float* __restrict__ c2, T* __restrict__ dx) {
    const long i = (long)blockIdx.x * BN_THREADS + threadIdx.x;
    if (i >= quads) return;
    const int tc = (int)(i % tc_n);
    const float4 mu = ld4(mean + 4 * tc), is = ld4(invstd + 4 * tc), sc = ld4(scale + 4 * tc);
    const float4 a = ld4(c1 + 4 * tc), b = ld4(c2 + 4 * tc);
    float4 g = ld4(dy + 4 * i);
    const float4 v = ld4(x + 4 * i);
    if (MODE == 1) {
        const float4 sh = ld4(shift + 4 * tc);
        if (!(bn_apply(v.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
        if (!(bn_apply(v.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
        if (!(bn_apply(v.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
        if (!(bn_apply(v.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
    }
    float4 o;
    o.x = sc.x * (g.x - a.x - (v.x - mu.x) * is.x * b.x);
    o.y = sc.y * (g.y - a.y - (v.y - mu.y) * is.y * b.y);
    o.z = sc.z * (g.z - a.z - (v.z - mu.z) * is.z * b.z);
    o.w = sc.w * (g.w - a.w - (v.w - mu.w) * is.w * b.w);
    st4(dx + 4 * i, o);
}

// ---- decoder conv units: y = LeakyReLU(conv(x) + bias) (image_encoder.py:141-193).  The library convolution is called
// without its bias; bias add and activation are ONE in-place pass, and the backward is one pass that writes the gradient of the
// pre-activation and the per-(block, channel) partial sums of the bias gradient, folded by the last-arriving workgroup
// (through ATen: a broadcast add, an in-place activation, an activation backward and a separate reduction over N*H*W).
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bias_leaky_fwd_kernel(T* __restrict__ y, const float* __restrict__ bias, long quads,
                                                                    int tc_n, float slope) {
    const long i = (long)blockIdx.x * BN_THREADS + threadIdx.x;
    if (i >= quads) return;
    const float4 b = ld4(bias + 4 * (int)(i % tc_n));
    float4 v = ld4(y + 4 * i);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    v.x = v.x > 0.f ? v.x : slope * v.x;
    v.y = v.y > 0.f ? v.y : slope * v.y;
    v.z = v.z > 0.f ? v.z : slope * v.z;
    v.w = v.w > 0.f ? v.w : slope * v.w;
    st4(y + 4 * i, v);
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bias_leaky_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, long R, int C,
                                                                    int tc_n, int rows_per_block, float slope, T* __restrict__ g_out,
                                                                    float* __restrict__ pa, float* __restrict__ pb,
                                                                    unsigned* __restrict__ ticket, float* __restrict__ dbias) {
    __shared__ float4 lds[2 * BN_THREADS];
    const int ri_n = BN_THREADS / tc_n;
    const int tc = threadIdx.x % tc_n, ri = threadIdx.x / tc_n;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(r0 + rows_per_block, R);
    float4 s = {0, 0, 0, 0};
    const float4 zero = {0, 0, 0, 0};
    for (long r = r0 + ri; r < r1; r += ri_n) {
        const size_t o = r * C + 4 * tc;
        float4 g = ld4(dy + o);
        const float4 out = ld4(y + o);               // slope > 0: the output has the sign of the pre-activation
        if (!(out.x > 0.f)) g.x *= slope;
        if (!(out.y > 0.f)) g.y *= slope;
        if (!(out.z > 0.f)) g.z *= slope;
        if (!(out.w > 0.f)) g.w *= slope;
        st4(g_out + o, g);
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
    }
    fold_rows(s, zero, tc_n, ri_n, pa, pb, C, lds);
    if (!dbias || !last_block_arrived(ticket)) return;
    double sa[4], sb[4];
    if (!fold_partials(pa, pb, (int)gridDim.x, C, tc_n, sa, sb, lds)) return;
#pragma unroll
    for (int i = 0; i < 4; i++) dbias[4 * tc + i] = (float)sa[i];
}


bool bad_shape(long R, int C) { return R <= 0 || C < 16 || C > 1024 || (C & (C - 1)) != 0; }

}  // namespace

extern "C" size_t scp_batchnorm_workspace(long R, int C) {
    if (bad_shape(R, C)) return 0;
    return ((size_t)2 * geometry(R, C).blocks * C + (size_t)4 * C) * sizeof(float);
}

namespace {

template <typename T>
int bn_forward_impl(const T* x, const T* skip, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    long long* batches_tracked, float momentum, float eps, long R, int C, int relu, int training, T* y,
                    float* save_mean, float* save_invstd, float* save_scale, float* save_shift, void* workspace,
                    size_t workspace_bytes, unsigned* ticket, void* stream) {
    if (bad_shape(R, C)) return scp::fail(hipErrorInvalidValue, "batchnorm: C must be a power of two in [16,1024], R > 0");
    if (!x || !y || !save_mean || !save_invstd || !save_scale || !save_shift || (training && !ticket))
        return scp::fail(hipErrorInvalidValue, "batchnorm: null argument");
    if (!training && (!running_mean || !running_var))
        return scp::fail(hipErrorInvalidValue, "batchnorm: eval mode needs running statistics");
    if (workspace_bytes < scp_batchnorm_workspace(R, C)) return scp::fail(hipErrorInvalidValue, "batchnorm: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geometry g = geometry(R, C);
    float* psum = static_cast<float*>(workspace);
    float* psq = psum + (size_t)g.blocks * C;
    const FwdFinalize fin{R, gamma, beta, running_mean, running_var, batches_tracked, momentum, eps, save_mean, save_invstd,
                          save_scale, save_shift};
    if (training) {
        hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(g.blocks), dim3(BN_THREADS), 0, st, x, R, C, g.tc, g.rows_per_block, psum, psq,
                           ticket, fin);
        if (int e = scp::check_launch("batchnorm stats")) return e;
    } else {
        hipLaunchKernelGGL(bn_eval_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, fin);
        if (int e = scp::check_launch("batchnorm finalize")) return e;
    }
    const long quads = R * C / 4;
    const dim3 grid((unsigned)((quads + BN_THREADS - 1) / BN_THREADS));
#define SCP_BN_APPLY(S, A) \
    hipLaunchKernelGGL((bn_apply_kernel<T, S, A>), grid, dim3(BN_THREADS), 0, st, x, skip, quads, g.tc, save_scale, save_shift, y)
    if (skip && relu) SCP_BN_APPLY(true, true);
    else if (skip) SCP_BN_APPLY(true, false);
    else if (relu) SCP_BN_APPLY(false, true);
    else SCP_BN_APPLY(false, false);
#undef SCP_BN_APPLY
    return scp::check_launch("batchnorm apply");
}

template <typename T>
int bn_backward_impl(const T* dy, const T* x, const T* y, const float* save_mean, const float* save_invstd,
                     const float* save_scale, const float* save_shift, long R, int C, int relu, int has_skip, int training, T* dx,
                     T* dskip, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, unsigned* ticket,
                     void* stream) {
    if (bad_shape(R, C)) return scp::fail(hipErrorInvalidValue, "batchnorm: C must be a power of two in [16,1024], R > 0");
    if (!dy || !x || !dx || !save_mean || !save_invstd || !save_scale || !save_shift || !ticket)
        return scp::fail(hipErrorInvalidValue, "batchnorm backward: null argument");
    const int mode = !relu ? 0 : (has_skip ? 2 : 1);
    if (mode == 2 && (!y || !dskip)) return scp::fail(hipErrorInvalidValue, "batchnorm backward: residual form needs y and dskip");
    if (workspace_bytes < scp_batchnorm_workspace(R, C)) return scp::fail(hipErrorInvalidValue, "batchnorm: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Geometry g = geometry(R, C);
    float* pg = static_cast<float*>(workspace);
    float* pgx = pg + (size_t)g.blocks * C;
    float* c1 = pgx + (size_t)g.blocks * C;
    float* c2 = c1 + C;
#define SCP_BN_REDUCE(M) \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, M>), dim3(g.blocks), dim3(BN_THREADS), 0, st, dy, x, y, R, C, g.tc, \
                       g.rows_per_block, save_mean, save_invstd, save_scale, save_shift, dskip, pg, pgx, ticket, training, c1, c2, \
                       dgamma, dbeta)
    if (mode == 0) SCP_BN_REDUCE(0);
    else if (mode == 1) SCP_BN_REDUCE(1);
    else SCP_BN_REDUCE(2);
#undef SCP_BN_REDUCE
    if (int e = scp::check_launch("batchnorm backward reduce")) return e;
    const long quads = R * C / 4;
    const dim3 grid((unsigned)((quads + BN_THREADS - 1) / BN_THREADS));
    // with a residual the masked gradient was written to dskip by the reduce pass: read that, no mask work
    const T* gsrc = mode == 2 ? dskip : dy;
    if (mode == 1)
        hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 1>), grid, dim3(BN_THREADS), 0, st, gsrc, x, quads, g.tc, save_mean, save_invstd,
                           save_scale, save_shift, c1, c2, dx);
    else
        hipLaunchKernelGGL((bn_bwd_dx_kernel<T, 0>), grid, dim3(BN_THREADS), 0, st, gsrc, x, quads, g.tc, save_mean, save_invstd,
                           save_scale, save_shift, c1, c2, dx);
    return scp::check_launch("batchnorm backward dx");
}

}  // namespace

extern "C" int scp_batchnorm_act_forward(const float* x, const float* skip, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, long long* batches_tracked,
                                         float momentum, float eps, long R, int C, int relu, int training, float* y,
                                         float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                                         void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream) {
    return bn_forward_impl<float>(x, skip, gamma, beta, running_mean, running_var, batches_tracked, momentum, eps, R, C, relu, training, y,
                                  save_mean, save_invstd, save_scale, save_shift, workspace, workspace_bytes, ticket, stream);
}

extern "C" int scp_batchnorm_apply(const float* x, const float* skip, const float* scale, const float* shift, long R, int C, int relu,
                                   float* y, void* stream) {
    if (bad_shape(R, C)) return scp::fail(hipErrorInvalidValue, "batchnorm: C must be a power of two in [16,1024], R > 0");
    if (!x || !y || !scale || !shift) return scp::fail(hipErrorInvalidValue, "batchnorm_apply: null argument");
    const long quads = R * C / 4;
    const dim3 grid((unsigned)((quads + BN_THREADS - 1) / BN_THREADS));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tc = C / 4;
#define SCP_BN_APPLY(S, A) hipLaunchKernelGGL((bn_apply_kernel<float, S, A>), grid, dim3(BN_THREADS), 0, st, x, skip, quads, tc, scale, shift, y)
    if (skip && relu) SCP_BN_APPLY(true, true);
    else if (skip) SCP_BN_APPLY(true, false);
    else if (relu) SCP_BN_APPLY(false, true);
    else SCP_BN_APPLY(false, false);
#undef SCP_BN_APPLY
    return scp::check_launch("batchnorm apply");
}

extern "C" int scp_batchnorm_act_backward(const float* dy, const float* x, const float* y, const float* save_mean,
                                          const float* save_invstd, const float* save_scale, const float* save_shift,
                                          long R, int C, int relu, int has_skip, int training, float* dx, float* dskip,
                                          float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                          unsigned* ticket, void* stream) {
    return bn_backward_impl<float>(dy, x, y, save_mean, save_invstd, save_scale, save_shift, R, C, relu, has_skip, training, dx, dskip,
                                   dgamma, dbeta, workspace, workspace_bytes, ticket, stream);
}

// bf16 activation storage (BASELINE configs[4] precision); parameters, statistics and workspace are fp32 as above
extern "C" int scp_batchnorm_act_forward_bf16(const void* x, const void* skip, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, long long* batches_tracked,
                                              float momentum, float eps, long R, int C, int relu, int training, void* y,
                                              float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                                              void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream) {
    return bn_forward_impl<__bf16>(static_cast<const __bf16*>(x), static_cast<const __bf16*>(skip), gamma, beta, running_mean, running_var,
                                   batches_tracked, momentum, eps, R, C, relu, training, static_cast<__bf16*>(y), save_mean,
                                   save_invstd, save_scale, save_shift, workspace, workspace_bytes, ticket, stream);
}

extern "C" int scp_batchnorm_act_backward_bf16(const void* dy, const void* x, const void* y, const float* save_mean,
                                               const float* save_invstd, const float* save_scale, const float* save_shift,
                                               long R, int C, int relu, int has_skip, int training, void* dx, void* dskip,
                                               float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                               unsigned* ticket, void* stream) {
    return bn_backward_impl<__bf16>(static_cast<const __bf16*>(dy), static_cast<const __bf16*>(x), static_cast<const __bf16*>(y), save_mean,
                                    save_invstd, save_scale, save_shift, R, C, relu, has_skip, training, static_cast<__bf16*>(dx),
                                    static_cast<__bf16*>(dskip), dgamma, dbeta, workspace, workspace_bytes, ticket, stream);
}

namespace {
template <typename T>
int bias_leaky_fwd(T* y, const float* bias, float slope, long R, int C, void* stream) {
    if (bad_shape(R, C)) return scp::fail(hipErrorInvalidValue, "bias_leaky: C must be a power of two in [16,1024], R > 0");
    if (!y || !bias) return scp::fail(hipErrorInvalidValue, "bias_leaky: null argument");
    const long quads = R * C / 4;
    hipLaunchKernelGGL(bias_leaky_fwd_kernel<T>, dim3((unsigned)((quads + BN_THREADS - 1) / BN_THREADS)), dim3(BN_THREADS), 0,
                       static_cast<hipStream_t>(stream), y, bias, quads, C / 4, slope);
    return scp::check_launch("bias_leaky forward");
}
template <typename T>
int bias_leaky_bwd(const T* dy, const T* y, float slope, long R, int C, T* g, float* dbias, void* workspace, size_t workspace_bytes,
                   unsigned* ticket, void* stream) {
    if (bad_shape(R, C)) return scp::fail(hipErrorInvalidValue, "bias_leaky: C must be a power of two in [16,1024], R > 0");
    if (!dy || !y || !g || (dbias && !ticket)) return scp::fail(hipErrorInvalidValue, "bias_leaky backward: null argument");
    if (workspace_bytes < scp_batchnorm_workspace(R, C)) return scp::fail(hipErrorInvalidValue, "bias_leaky: workspace too small");
    const Geometry ge = geometry(R, C);
    float* pa = static_cast<float*>(workspace);
    float* pb = pa + (size_t)ge.blocks * C;
    hipLaunchKernelGGL(bias_leaky_bwd_kernel<T>, dim3(ge.blocks), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), dy, y, R, C, ge.tc,
                       ge.rows_per_block, slope, g, pa, pb, ticket, dbias);
    return scp::check_launch("bias_leaky backward");
}
}  // namespace

extern "C" int scp_bias_leaky_relu_forward(float* y, const float* bias, float slope, long R, int C, void* stream) {
    return bias_leaky_fwd<float>(y, bias, slope, R, C, stream);
}
extern "C" int scp_bias_leaky_relu_forward_bf16(void* y, const float* bias, float slope, long R, int C, void* stream) {
    return bias_leaky_fwd<__bf16>(static_cast<__bf16*>(y), bias, slope, R, C, stream);
}
extern "C" int scp_bias_leaky_relu_backward(const float* dy, const float* y, float slope, long R, int C, float* g, float* dbias,
                                            void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream) {
    return bias_leaky_bwd<float>(dy, y, slope, R, C, g, dbias, workspace, workspace_bytes, ticket, stream);
}
extern "C" int scp_bias_leaky_relu_backward_bf16(const void* dy, const void* y, float slope, long R, int C, void* g, float* dbias,
                                                 void* workspace, size_t workspace_bytes, unsigned* ticket, void* stream) {
    return bias_leaky_bwd<__bf16>(static_cast<const __bf16*>(dy), static_cast<const __bf16*>(y), slope, R, C, static_cast<__bf16*>(g), dbias,
                                  workspace, workspace_bytes, ticket, stream);
}
