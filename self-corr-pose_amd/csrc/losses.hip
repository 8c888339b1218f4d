// self-corr-pose_amd/csrc/losses.hip -- the per-pixel image losses of the depth render group as two forward and three backward
// launches instead of ~150 small ATen kernels: they sit on the step's serial correspondence / render / loss chain, where one
// latency-bound kernel is resident at a time and the matrix cores idle.
//
// Semantics (model/util/loss_utils.py, restated in scp_amd/losses.py):
//   compute_mask_loss  :236-244  0.2 * mean over (H,W) of sum_{l=0..4} (area-pool_l along W of (mask_pred - mask))^2
//                                (the "pyramid" pools along W only because 3-D tensors are fed to F.interpolate)
//   compute_depth_loss :273-284  keep = (mask * depth_mask != 0) & (depth != 0); batch-global
//                                depth_scale = mean_{depth_mask != 0}(depth_pred) / mean_{mask*depth != 0}(depth);
//                                diff = keep ? depth_pred - depth_scale * depth : 0;  loss = mean(1 - relu(1 - diff^2))
//   compute_match_loss :317-320  mean(||match - match_gt||_2 * [(match_mask > 0) & (mask > 0)])
// Inputs as the step has them: depth_out [B,4,H,W] is the depth render (plane 2 = depth, plane 3 = alpha; alpha is mask_pred AND
// depth_mask: renderer.py takes the mask from the depth pass), match_out [B,4,H,W] the canonical-xyz render (planes 0..2 = match_gt,
// plane 3 = match_mask), match [B,3,H,W] the predicted correspondence, mask / depth [B,H,W] the data.
// One workgroup per image row (W threads, W a power of two in [32, 1024]); row partial sums are folded by the caller (three tiny
// reductions) -- deterministic, no atomics.  The gradient of depth_scale w.r.t. depth_pred (a batch-global coupling) is the third
// backward launch.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// block-wide sum of up to 4 values per thread; result valid in thread 0
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* lds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = wave_sum(v[i]);
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < N; i++) lds[wave * N + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            float s = lds[i];
            for (int w = 1; w < nw; w++) s += lds[w * N + i];
            v[i] = s;
        }
    }
    __syncthreads();
}

// sums for depth_scale: out[blockIdx][4] = (sum depth_pred*m_pred, sum m_pred, sum depth*m_gt, sum m_gt) over this block's rows
__global__ void depth_sums_kernel(const float* __restrict__ depth_out, const float* __restrict__ depth, const float* __restrict__ mask,
                                  int rows, int H, int W, float* __restrict__ out) {
    __shared__ float lds[16 * 4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int b = r / H, h = r - b * H;
        const size_t plane = (size_t)H * W, px = (size_t)h * W + threadIdx.x;
        const float a = depth_out[((size_t)b * 4 + 3) * plane + px], dp = depth_out[((size_t)b * 4 + 2) * plane + px];
        const float dg = depth[(size_t)b * plane + px], m = mask[(size_t)b * plane + px];
        const float mp = a != 0.f ? 1.f : 0.f, mg = (m * dg) != 0.f ? 1.f : 0.f;
        v[0] += dp * mp; v[1] += mp; v[2] += dg * mg; v[3] += mg;
    }
    block_sum(v, lds);
    if (threadIdx.x == 0)
#pragma unroll
        for (int i = 0; i < 4; i++) out[blockIdx.x * 4 + i] = v[i];
}

// every workgroup folds the nparts x 4 partial sums itself (fixed order): (depth_scale, 1 / (N1 * mean_gt))
__device__ __forceinline__ void fold_scale(const float* __restrict__ parts, int nparts, float* lds, float& scale, float& coef) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nparts; i += blockDim.x)
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] += parts[i * 4 + k];
    block_sum(v, lds);
    if (threadIdx.x == 0) {
        const float mean_gt = v[2] / v[3];
        lds[0] = (v[0] / v[1]) / mean_gt;
        lds[1] = 1.f / (v[1] * mean_gt);
    }
    __syncthreads();
    scale = lds[0];
    coef = lds[1];
    __syncthreads();
}

struct Pixel { float a, dp, dg, m, mm, v[3]; };
__device__ __forceinline__ Pixel load_pixel(const float* depth_out, const float* depth, const float* mask, const float* match,
                                            const float* match_out, int b, int h, int H, int W) {
    const size_t plane = (size_t)H * W, px = (size_t)h * W + threadIdx.x;
    Pixel p;
    p.a = depth_out[((size_t)b * 4 + 3) * plane + px];
    p.dp = depth_out[((size_t)b * 4 + 2) * plane + px];
    p.dg = depth[(size_t)b * plane + px];
    p.m = mask[(size_t)b * plane + px];
    p.mm = match_out[((size_t)b * 4 + 3) * plane + px];
#pragma unroll
    for (int c = 0; c < 3; c++) p.v[c] = match[((size_t)b * 3 + c) * plane + px] - match_out[((size_t)b * 4 + c) * plane + px];
    return p;
}

// the five pyramid levels of d = mask_pred - mask at this pixel: d_l = mean of its aligned group of 2^l pixels along W
// (cascade of pair means, the order scp_amd/losses.py uses)
__device__ __forceinline__ void pyramid(float d0, float (&d)[5]) {
    d[0] = d0;
#pragma unroll
    for (int l = 1; l < 5; l++) d[l] = (d[l - 1] + __shfl_xor(d[l - 1], 1 << (l - 1))) * 0.5f;
}

// rowsum[row][3] = per-row means (mask pyramid term, depth term, match term): loss_k[b] = c_k * mean_h rowsum[b][h][k]
__global__ void image_losses_forward_kernel(const float* __restrict__ depth_out, const float* __restrict__ depth,
                                            const float* __restrict__ mask, const float* __restrict__ match,
                                            const float* __restrict__ match_out,
                                            const float* __restrict__ parts, int nparts, int H, int W, float* __restrict__ rowsum) {
    __shared__ float lds[16 * 4];
    float scale, coef;
    fold_scale(parts, nparts, lds, scale, coef);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const Pixel p = load_pixel(depth_out, depth, mask, match, match_out, b, h, H, W);
    float d[5];
    pyramid(p.a - p.m, d);
    float v[3];
    v[0] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3] + d[4] * d[4];
    const bool keep = (p.m * p.a) != 0.f && p.dg != 0.f;
    const float diff = keep ? p.dp - scale * p.dg : 0.f;
    const float sq = diff * diff;
    v[1] = 1.f - fmaxf(1.f - sq, 0.f);
    const bool valid = p.mm > 0.f && p.m > 0.f;
    v[2] = valid ? sqrtf(p.v[0] * p.v[0] + p.v[1] * p.v[1] + p.v[2] * p.v[2]) : 0.f;
    block_sum(v, lds);
    if (threadIdx.x == 0) {
        const float inv_w = 1.f / (float)W;
        rowsum[blockIdx.x * 3 + 0] = v[0] * inv_w;
        rowsum[blockIdx.x * 3 + 1] = v[1] * inv_w;
        rowsum[blockIdx.x * 3 + 2] = v[2] * inv_w;
    }
}

// gradients for upstream per-image gradients g_mask / g_depth / g_match [B] (of the three loss vectors):
//   grad_depth_out [B,4,H,W] (plane 3 = mask term, plane 2 = local depth term, planes 0,1 = 0), grad_match [B,3,H,W],
//   gsum[row] = sum over the row of (local depth gradient * depth) -- the coupling through depth_scale, applied by the next launch
__global__ void image_losses_backward_kernel(const float* __restrict__ depth_out, const float* __restrict__ depth,
                                             const float* __restrict__ mask, const float* __restrict__ match,
                                             const float* __restrict__ match_out,
                                             const float* __restrict__ parts, int nparts, const float* __restrict__ g_mask,
                                             const float* __restrict__ g_depth, const float* __restrict__ g_match, int H, int W,
                                             float* __restrict__ grad_depth_out, float* __restrict__ grad_match,
                                             float* __restrict__ gsum) {
    __shared__ float lds[16 * 4];
    float scale, coef;
    fold_scale(parts, nparts, lds, scale, coef);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const Pixel p = load_pixel(depth_out, depth, mask, match, match_out, b, h, H, W);
    const size_t plane = (size_t)H * W, px = (size_t)h * W + threadIdx.x;
    const float inv_hw = 1.f / ((float)H * (float)W);
    float d[5];
    pyramid(p.a - p.m, d);
    // loss_mask[b] = 0.2 / (H W) sum_{h,x} sum_l d_l[x]^2 and d sum_l d_l^2 / d a[x'] summed over the group = 2 sum_l d_l[x']
    const float ga = g_mask[b] * 0.2f * inv_hw * 2.f * (d[0] + d[1] + d[2] + d[3] + d[4]);
    const bool keep = (p.m * p.a) != 0.f && p.dg != 0.f;
    const float diff = keep ? p.dp - scale * p.dg : 0.f;
    const float gd = (keep && diff * diff < 1.f) ? g_depth[b] * inv_hw * 2.f * diff : 0.f;
    grad_depth_out[((size_t)b * 4 + 0) * plane + px] = 0.f;
    grad_depth_out[((size_t)b * 4 + 1) * plane + px] = 0.f;
    grad_depth_out[((size_t)b * 4 + 2) * plane + px] = gd;
    grad_depth_out[((size_t)b * 4 + 3) * plane + px] = ga;
    const bool valid = p.mm > 0.f && p.m > 0.f;
    const float e = sqrtf(p.v[0] * p.v[0] + p.v[1] * p.v[1] + p.v[2] * p.v[2]);
    const float gm = (valid && e > 0.f) ? g_match[b] * inv_hw / e : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) grad_match[((size_t)b * 3 + c) * plane + px] = gm * p.v[c];
    float v[1] = {gd * p.dg};
    block_sum(v, lds);
    if (threadIdx.x == 0) gsum[blockIdx.x] = v[0];
}

// depth_pred += -(sum_j gd_j depth_j) * d depth_scale / d depth_pred = -G * [alpha != 0] / (N1 mean_gt)
__global__ void depth_scale_backward_kernel(const float* __restrict__ depth_out, const float* __restrict__ parts, int nparts,
                                            const float* __restrict__ G, int H, int W, float* __restrict__ grad_depth_out) {
    __shared__ float lds[16 * 4];
    float scale, coef;
    fold_scale(parts, nparts, lds, scale, coef);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t plane = (size_t)H * W, px = (size_t)h * W + threadIdx.x;
    const float a = depth_out[((size_t)b * 4 + 3) * plane + px];
    if (a != 0.f) grad_depth_out[((size_t)b * 4 + 2) * plane + px] -= G[0] * coef;
}

// compute_texture_loss (loss_utils.py:246-252): fg = mask > 0, img_gt = img * fg, white = 1 - fg + img_gt,
//   per pixel 0.75 sum_c (img_gt_c - tex_c * alpha)^2 + mean_c |white_c - tex_c|;  tex_out [B,4,H,W] = (rgb, alpha) of the soft-texture pass
template <bool BACKWARD>
__global__ void texture_loss_kernel(const float* __restrict__ tex_out, const float* __restrict__ img, const float* __restrict__ mask,
                                    const float* __restrict__ g_tex, int H, int W, float* __restrict__ rowsum,
                                    float* __restrict__ grad_tex_out) {
    __shared__ float lds[16];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const size_t plane = (size_t)H * W, px = (size_t)h * W + threadIdx.x;
    const float fg = mask[(size_t)b * plane + px] > 0.f ? 1.f : 0.f;
    const float a = tex_out[((size_t)b * 4 + 3) * plane + px];
    float sq = 0.f, ab = 0.f, ga = 0.f;
    const float g = BACKWARD ? g_tex[b] / ((float)H * (float)W) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = tex_out[((size_t)b * 4 + c) * plane + px];
        const float gt = img[((size_t)b * 3 + c) * plane + px] * fg;
        const float white = 1.f - fg + gt;
        const float r = gt - t * a, w = white - t;
        if (BACKWARD) {
            const float sgn = w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f);
            grad_tex_out[((size_t)b * 4 + c) * plane + px] = g * (-1.5f * r * a - sgn * (1.f / 3.f));
            ga += -1.5f * r * t;
        } else {
            sq += r * r;
            ab += fabsf(w);
        }
    }
    if (BACKWARD) {
        grad_tex_out[((size_t)b * 4 + 3) * plane + px] = g * ga;
    } else {
        float v[1] = {0.75f * sq + ab * (1.f / 3.f)};
        block_sum(v, lds);
        if (threadIdx.x == 0) rowsum[blockIdx.x] = v[0] / (float)W;
    }
}

bool bad(int B, int H, int W) { return B <= 0 || H <= 0 || W < 32 || W > 1024 || (W & (W - 1)); }

}  // namespace

extern "C" int scp_image_losses_parts(void) { return 256; }

extern "C" int scp_image_losses_forward(const float* depth_out, const float* depth, const float* mask, const float* match,
                                        const float* match_out, int B, int H, int W, float* parts,
                                        float* rowsum, void* stream) {
    if (bad(B, H, W)) return scp::fail(hipErrorInvalidValue, "image_losses: W must be a power of two in [32,1024]");
    if (!depth_out || !depth || !mask || !match || !match_out || !parts || !rowsum)
        return scp::fail(hipErrorInvalidValue, "image_losses: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nparts = scp_image_losses_parts();
    hipLaunchKernelGGL(depth_sums_kernel, dim3(nparts), dim3(W), 0, st, depth_out, depth, mask, B * H, H, W, parts);
    hipLaunchKernelGGL(image_losses_forward_kernel, dim3(B * H), dim3(W), 0, st, depth_out, depth, mask, match, match_out,
                       parts, nparts, H, W, rowsum);
    return scp::check_launch("image_losses_forward");
}

extern "C" int scp_image_losses_backward(const float* depth_out, const float* depth, const float* mask, const float* match,
                                         const float* match_out, const float* parts, const float* g_mask,
                                         const float* g_depth, const float* g_match, int B, int H, int W, float* grad_depth_out,
                                         float* grad_match, float* gsum, void* stream) {
    if (bad(B, H, W)) return scp::fail(hipErrorInvalidValue, "image_losses: W must be a power of two in [32,1024]");
    if (!depth_out || !parts || !g_mask || !g_depth || !g_match || !grad_depth_out || !grad_match || !gsum)
        return scp::fail(hipErrorInvalidValue, "image_losses_backward: null argument");
    hipLaunchKernelGGL(image_losses_backward_kernel, dim3(B * H), dim3(W), 0, static_cast<hipStream_t>(stream), depth_out, depth, mask,
                       match, match_out, parts, scp_image_losses_parts(), g_mask, g_depth, g_match, H, W, grad_depth_out,
                       grad_match, gsum);
    return scp::check_launch("image_losses_backward");
}

extern "C" int scp_image_losses_backward_scale(const float* depth_out, const float* parts, const float* G, int B, int H, int W,
                                               float* grad_depth_out, void* stream) {
    if (bad(B, H, W)) return scp::fail(hipErrorInvalidValue, "image_losses: W must be a power of two in [32,1024]");
    if (!depth_out || !parts || !G || !grad_depth_out) return scp::fail(hipErrorInvalidValue, "image_losses_backward_scale: null argument");
    hipLaunchKernelGGL(depth_scale_backward_kernel, dim3(B * H), dim3(W), 0, static_cast<hipStream_t>(stream), depth_out, parts,
                       scp_image_losses_parts(), G, H, W, grad_depth_out);
    return scp::check_launch("image_losses_backward_scale");
}

extern "C" int scp_texture_loss_forward(const float* tex_out, const float* img, const float* mask, int B, int H, int W, float* rowsum,
                                        void* stream) {
    if (bad(B, H, W)) return scp::fail(hipErrorInvalidValue, "texture_loss: W must be a power of two in [32,1024]");
    if (!tex_out || !img || !mask || !rowsum) return scp::fail(hipErrorInvalidValue, "texture_loss: null argument");
    hipLaunchKernelGGL(texture_loss_kernel<false>, dim3(B * H), dim3(W), 0, static_cast<hipStream_t>(stream), tex_out, img, mask,
                       (const float*)nullptr, H, W, rowsum, (float*)nullptr);
    return scp::check_launch("texture_loss_forward");
}

extern "C" int scp_texture_loss_backward(const float* tex_out, const float* img, const float* mask, const float* g_tex, int B, int H,
                                         int W, float* grad_tex_out, void* stream) {
    if (bad(B, H, W)) return scp::fail(hipErrorInvalidValue, "texture_loss: W must be a power of two in [32,1024]");
    if (!tex_out || !img || !mask || !g_tex || !grad_tex_out) return scp::fail(hipErrorInvalidValue, "texture_loss_backward: null argument");
    hipLaunchKernelGGL(texture_loss_kernel<true>, dim3(B * H), dim3(W), 0, static_cast<hipStream_t>(stream), tex_out, img, mask, g_tex,
                       H, W, (float*)nullptr, grad_tex_out);
    return scp::check_launch("texture_loss_backward");
}
