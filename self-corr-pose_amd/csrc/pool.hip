// self-corr-pose_amd/csrc/pool.hip -- the ResNet stem's MaxPool2d(kernel 3, stride 2, padding 1) on NHWC activations, forward
// and backward (torchvision resnet18 as instantiated by model/module/network/image_encoder.py:119-139: conv1 -> bn1 -> relu ->
// maxpool).
//
// ATen's NHWC kernels run this at 2.1 TB/s forward (78 us for [32,64,128,128]) and write / re-read an int64 index tensor for
// the backward (113 us).  Here: forward = one thread per (output pixel, 4 channels), 9 coalesced float4 reads, the position of
// the maximum inside its window (0..8) kept as ONE BYTE per element; backward = gather form, one thread per (input pixel,
// 4 channels) over the <= 4 windows that contain it -- no atomics, no zero-fill pass.  Semantics are ATen's: the window is
// scanned row by row and a later element replaces the maximum only if it is strictly greater (or NaN), so among equal values
// (the zeros a ReLU leaves) the first one takes the gradient.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

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

__device__ __forceinline__ void take(float v, int k, float& best, int& at) {
    if (v > best || v != v) { best = v; at = k; }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uchar4* __restrict__ where,
                                                          int N, int H, int W, int C4) {
    const int OH = H / 2, OW = W / 2;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)N * OH * OW * C4) return;
    const int c4 = (int)(id % C4);
    long p = id / C4;
    const int ow = (int)(p % OW);
    p /= OW;
    const int oh = (int)(p % OH);
    const int n = (int)(p / OH);
    const size_t base = (size_t)n * H * W * C4 + c4;
    float4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int ax = -1, ay = -1, az = -1, aw = -1;
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
        const int ih = 2 * oh - 1 + kh;
        if (ih < 0 || ih >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; kw++) {
            const int iw = 2 * ow - 1 + kw;
            if (iw < 0 || iw >= W) continue;
            const float4 v = load4(x, base + ((size_t)ih * W + iw) * C4);
            const int k = 3 * kh + kw;
            if (ax < 0) { best = v; ax = ay = az = aw = k; continue; }      // first element of the window (ATen starts from it)
            take(v.x, k, best.x, ax); take(v.y, k, best.y, ay); take(v.z, k, best.z, az); take(v.w, k, best.w, aw);
        }
    }
    store4(y, (size_t)id, best);
    where[id] = make_uchar4((unsigned char)ax, (unsigned char)ay, (unsigned char)az, (unsigned char)aw);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uchar4* __restrict__ where,
                                                          T* __restrict__ dx, int N, int H, int W, int C4) {
    const int OH = H / 2, OW = W / 2;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)N * H * W * C4) return;
    const int c4 = (int)(id % C4);
    long p = id / C4;
    const int iw = (int)(p % W);
    p /= W;
    const int ih = (int)(p % H);
    const int n = (int)(p / H);
    const size_t obase = (size_t)n * OH * OW * C4 + c4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    // windows that contain (ih, iw): oh in {ih / 2, (ih + 1) / 2} (equal for even ih), same along w
    const int oh0 = ih / 2, oh1 = (ih + 1) / 2, ow0 = iw / 2, ow1 = (iw + 1) / 2;
    for (int a = 0; a < 2; a++) {
        const int oh = a ? oh1 : oh0;
        if ((a && oh1 == oh0) || oh >= OH) continue;
        const int kh = ih - (2 * oh - 1);
        for (int b = 0; b < 2; b++) {
            const int ow = b ? ow1 : ow0;
            if ((b && ow1 == ow0) || ow >= OW) continue;
            const int k = 3 * kh + (iw - (2 * ow - 1));
            const size_t o = obase + ((size_t)oh * OW + ow) * C4;
            const uchar4 at = where[o];
            const float4 g = load4(dy, o);
            if (at.x == k) acc.x += g.x;
            if (at.y == k) acc.y += g.y;
            if (at.z == k) acc.z += g.z;
            if (at.w == k) acc.w += g.w;
        }
    }
    store4(dx, (size_t)id, acc);
}

int check(int N, int H, int W, int C, const void* a, const void* b, const void* c) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return scp::fail(hipErrorInvalidValue, "maxpool3x3s2: empty problem");
    if ((H & 1) || (W & 1) || (C & 3)) return scp::fail(hipErrorInvalidValue, "maxpool3x3s2: H, W must be even and C a multiple of 4");
    if (!a || !b || !c) return scp::fail(hipErrorInvalidValue, "maxpool3x3s2: null argument");
    return 0;
}

template <typename T>
int fwd(const void* x, void* y, unsigned char* where, int N, int H, int W, int C, void* stream) {
    if (int e = check(N, H, W, C, x, y, where)) return e;
    const long total = (long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const T*>(x), static_cast<T*>(y), reinterpret_cast<uchar4*>(where), N, H, W, C / 4);
    return scp::check_launch("maxpool3x3s2 forward");
}

template <typename T>
int bwd(const void* dy, const unsigned char* where, void* dx, int N, int H, int W, int C, void* stream) {
    if (int e = check(N, H, W, C, dy, where, dx)) return e;
    const long total = (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const T*>(dy), reinterpret_cast<const uchar4*>(where), static_cast<T*>(dx), N, H, W, C / 4);
    return scp::check_launch("maxpool3x3s2 backward");
}

}  // namespace

extern "C" int scp_maxpool3x3s2_forward(const float* x, float* y, unsigned char* where, int N, int H, int W, int C, void* stream) {
    return fwd<float>(x, y, where, N, H, W, C, stream);
}
extern "C" int scp_maxpool3x3s2_forward_bf16(const void* x, void* y, unsigned char* where, int N, int H, int W, int C, void* stream) {
    return fwd<__bf16>(x, y, where, N, H, W, C, stream);
}
extern "C" int scp_maxpool3x3s2_backward(const float* dy, const unsigned char* where, float* dx, int N, int H, int W, int C,
                                         void* stream) {
    return bwd<float>(dy, where, dx, N, H, W, C, stream);
}
extern "C" int scp_maxpool3x3s2_backward_bf16(const void* dy, const unsigned char* where, void* dx, int N, int H, int W, int C,
                                              void* stream) {
    return bwd<__bf16>(dy, where, dx, N, H, W, C, stream);
}
