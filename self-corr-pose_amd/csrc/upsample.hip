// self-corr-pose_amd/csrc/upsample.hip -- the decoder's exact-2x bilinear upsampling, NHWC fp32 (bf16 storage variant): forward
// and backward.
//
// Replaces the autograd backward of `F.interpolate(x, size=2x, mode="bilinear", align_corners=False)` in ResNet_Decoder._up
// (model/module/network/image_encoder.py:141-193 upsamples c5->c4, c4->c3, c3->c2 resolution).  ATen's NHWC backward
// scatters every output gradient into its four sources with atomicAdd (0.13 ms per call at [32,128,64,64], six calls per
// step, and a summation order that changes from run to run).  For an exact factor of two the map is separable with fixed
// taps: output 2k reads inputs (k-1, k) with weights (1/4, 3/4), output 2k+1 reads (k, k+1) with (3/4, 1/4), clamped at the
// borders.  So input pixel i GATHERS from outputs 2i-1, 2i, 2i+1, 2i+2 with weights (1/4, 3/4, 3/4, 1/4) (border taps
// folded onto the edge pixel) -- one thread per (input pixel, 4 channels), 16 coalesced float4 reads, no atomics,
// deterministic.  HBM-bound: reads the output gradient once (L2 serves the 4x overlap), writes the input gradient once.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

// taps of input index i along one axis of length n (output length 2n): up to 4 (output index, weight) pairs
__device__ __forceinline__ int taps(int i, int n, int idx[4], float w[4]) {
    int c = 0;
    // output o = 2k (+0/1): sources (i0, i1) = even: (k-1, k) w (0.25, 0.75); odd: (k, k+1) w (0.75, 0.25); i0 clamped at 0
    // (then all weight goes to input 0), i1 clamped at n-1
    if (i > 0) { idx[c] = 2 * i - 1; w[c] = 0.25f; c++; }                 // odd output 2(i-1)+1, upper source
    idx[c] = 2 * i; w[c] = (i == 0) ? 1.0f : 0.75f; c++;                  // even output 2i: lower source i-1 clamps onto 0 when i == 0
    idx[c] = 2 * i + 1; w[c] = (i == n - 1) ? 1.0f : 0.75f; c++;          // odd output 2i+1: upper source i+1 clamps onto n-1
    if (i < n - 1) { idx[c] = 2 * i + 2; w[c] = 0.25f; c++; }             // even output 2(i+1), lower source
    return c;
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4(const float* p, size_t quad) { return reinterpret_cast<const float4*>(p)[quad]; }
__device__ __forceinline__ float4 load4(const __bf16* p, size_t quad) {
    const bf16x4 v = reinterpret_cast<const bf16x4*>(p)[quad];
    return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float* p, size_t quad, float4 v) { reinterpret_cast<float4*>(p)[quad] = v; }
__device__ __forceinline__ void store4(__bf16* p, size_t quad, float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    reinterpret_cast<bf16x4*>(p)[quad] = o;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ gout, T* __restrict__ gin, int N,
                                                             int H, int W, int C4) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * H * W * C4;
    if (id >= total) return;
    const int c4 = (int)(id % C4);
    long p = id / C4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    int iy[4], ix[4];
    float wy[4], wx[4];
    const int ny = taps(y, H, iy, wy), nx = taps(x, W, ix, wx);
    const int OW = 2 * W;
    const size_t src = (size_t)n * (2 * H) * OW * C4 + c4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; a++) {
        float4 row = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < nx; b++) {
            const float4 v = load4(gout, src + ((size_t)iy[a] * OW + ix[b]) * C4);
            row.x += wx[b] * v.x; row.y += wx[b] * v.y; row.z += wx[b] * v.z; row.w += wx[b] * v.w;
        }
        acc.x += wy[a] * row.x; acc.y += wy[a] * row.y; acc.z += wy[a] * row.z; acc.w += wy[a] * row.w;
    }
    store4(gin, (size_t)id, acc);
}

// forward: one thread per (output pixel, 4 channels).  Output o = 2k + b reads inputs (k - 1 + b, k + b) clamped to the image,
// weights (1/4, 3/4) for b = 0 and (3/4, 1/4) for b = 1 -- ATen's align_corners=False arithmetic for an exact factor of two
// (source coordinate max((o + 0.5) / 2 - 0.5, 0), its integer part and fraction).  ATen's NHWC kernel reaches 0.7 TB/s at the
// decoder's sizes (120 us for [32,128,32,32] -> 64x64); this one is bound by the 4x larger write.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W,
                                                             int C4) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int OH = 2 * H, OW = 2 * W;
    const long total = (long)N * OH * OW * C4;
    if (id >= total) return;
    const int c4 = (int)(id % C4);
    long p = id / C4;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    // source coordinate s = max(o / 2 - 0.25, 0): integer part i0, fraction l1 (0.75 for even o > 0, 0.25 for odd o, 0 at o = 0)
    const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
    const size_t base = (size_t)n * H * W * C4 + c4;
    const float4 v00 = load4(in, base + ((size_t)y0 * W + x0) * C4), v01 = load4(in, base + ((size_t)y0 * W + x1) * C4);
    const float4 v10 = load4(in, base + ((size_t)y1 * W + x0) * C4), v11 = load4(in, base + ((size_t)y1 * W + x1) * C4);
    float4 o;
    o.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    o.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    o.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    o.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    store4(out, (size_t)id, o);
}

}  // namespace

namespace {
template <typename T>
int launch_upsample_bwd(const void* grad_out, void* grad_in, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return scp::fail(hipErrorInvalidValue, "upsample2x_backward: empty problem");
    if (C % 4 != 0) return scp::fail(hipErrorInvalidValue, "upsample2x_backward: C must be a multiple of 4");
    if (!grad_out || !grad_in) return scp::fail(hipErrorInvalidValue, "upsample2x_backward: null argument");
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const T*>(grad_out), static_cast<T*>(grad_in), N, H, W, C / 4);
    return scp::check_launch("upsample2x_backward");
}
}  // namespace

namespace {
template <typename T>
int launch_upsample_fwd(const void* in, void* out, int N, int H, int W, int C, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return scp::fail(hipErrorInvalidValue, "upsample2x_forward: empty problem");
    if (C % 4 != 0) return scp::fail(hipErrorInvalidValue, "upsample2x_forward: C must be a multiple of 4");
    if (!in || !out) return scp::fail(hipErrorInvalidValue, "upsample2x_forward: null argument");
    const long total = (long)N * (2 * H) * (2 * W) * (C / 4);
    hipLaunchKernelGGL(upsample2x_fwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const T*>(in), static_cast<T*>(out), N, H, W, C / 4);
    return scp::check_launch("upsample2x_forward");
}
}  // namespace

extern "C" int scp_upsample2x_bilinear_forward(const float* in, float* out, int N, int H, int W, int C, void* stream) {
    return launch_upsample_fwd<float>(in, out, N, H, W, C, stream);
}

extern "C" int scp_upsample2x_bilinear_forward_bf16(const void* in, void* out, int N, int H, int W, int C, void* stream) {
    return launch_upsample_fwd<__bf16>(in, out, N, H, W, C, stream);
}

extern "C" int scp_upsample2x_bilinear_backward(const float* grad_out, float* grad_in, int N, int H, int W, int C, void* stream) {
    return launch_upsample_bwd<float>(grad_out, grad_in, N, H, W, C, stream);
}

extern "C" int scp_upsample2x_bilinear_backward_bf16(const void* grad_out, void* grad_in, int N, int H, int W, int C, void* stream) {
    return launch_upsample_bwd<__bf16>(grad_out, grad_in, N, H, W, C, stream);
}
