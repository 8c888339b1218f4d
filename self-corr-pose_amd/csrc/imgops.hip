// self-corr-pose_amd/csrc/imgops.hip -- the encoder's input transform as ONE streaming pass:
// ColorJitter (brightness / contrast / saturation / hue in this call's random order) + Normalize, written
// straight into the NHWC layout the MIOpen convolutions want.
//
// Replaces model/module/encoder.py:18-19,31 (`resnet_transform(random_jitter(img))`, torchvision 0.11
// tensor ops, un-vendored).  Composed from torch ops this is ~60 elementwise launches over [B,3,H,W]
// planes, twice per step (the rotation-cycle branch encodes the rotated image as well): 2.3 ms of GPU
// time each.  The arithmetic is per pixel except the contrast anchor (per-image mean of the grey image
// at the point of the chain where contrast sits), so: one reduction kernel (skipped when contrast is
// off) + one apply kernel.  HBM-bound: 12 B/pixel in, 12 B/pixel out (+12 B/pixel for the reduction).
// Built with -ffp-contract=off and the same operation order as the published torchvision formulas
// (scp_amd/imgops.py restates them in torch), so results agree with that composition to rounding of the
// mean only.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr int JT_THREADS = 256;
constexpr int JT_PARTS = 64;  // reduction blocks per image

struct JitterArgs {
    const float* img;  // [N,3,H,W]
    float* out;        // [N,H,W,3] (nhwc != 0) or [N,3,H,W]
    float* partial;    // [N,JT_PARTS]
    int N, HW, nhwc;
    int order[4];      // op ids: 0 brightness, 1 contrast, 2 saturation, 3 hue; -1 = disabled slot
    float ratio[3], one_minus[3];
    float hue_shift;
    float mean[3], stdv[3];
};

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ float gray(float r, float g, float b) { return (0.2989f * r + 0.587f * g) + 0.114f * b; }

__device__ __forceinline__ void hue_op(float& r, float& g, float& b, float shift) {
    const float maxc = fmaxf(fmaxf(r, g), b), minc = fminf(fminf(r, g), b);
    const bool eqc = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eqc ? 1.f : maxc);
    const float crd = eqc ? 1.f : cr;
    const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
    const float hr = (maxc == r) ? (bc - gc) : 0.f;
    const float hg = (maxc == g && maxc != r) ? ((2.f + rc) - bc) : 0.f;
    const float hb = (maxc != g && maxc != r) ? ((4.f + gc) - rc) : 0.f;
    float h = fmodf(((hr + hg) + hb) / 6.f + 1.f, 1.f);
    h = fmodf(h + shift, 1.f);
    if (h < 0.f) h += 1.f;  // python-style remainder
    const float v = maxc;
    const float h6 = h * 6.f;
    const float fl = floorf(h6);
    const float f = h6 - fl;
    const int i = ((int)fl) % 6;
    const float p = clamp01(v * (1.f - s));
    const float q = clamp01(v * (1.f - s * f));
    const float t = clamp01(v * (1.f - s * (1.f - f)));
    switch (i) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// applies chain slots [first, last); `cmean` is the contrast anchor (only read by op 1)
__device__ __forceinline__ void chain(const JitterArgs& a, int first, int last, float cmean, float& r, float& g,
                                      float& b) {
    for (int k = first; k < last; k++) {
        const int op = a.order[k];
        if (op == 0) {
            r = clamp01(a.ratio[0] * r); g = clamp01(a.ratio[0] * g); b = clamp01(a.ratio[0] * b);
        } else if (op == 1) {
            const float t = a.one_minus[1] * cmean;
            r = clamp01(a.ratio[1] * r + t); g = clamp01(a.ratio[1] * g + t); b = clamp01(a.ratio[1] * b + t);
        } else if (op == 2) {
            const float t = a.one_minus[2] * gray(r, g, b);
            r = clamp01(a.ratio[2] * r + t); g = clamp01(a.ratio[2] * g + t); b = clamp01(a.ratio[2] * b + t);
        } else if (op == 3) {
            hue_op(r, g, b, a.hue_shift);
        }
    }
}

__device__ __forceinline__ int contrast_slot(const JitterArgs& a) {
    for (int k = 0; k < 4; k++)
        if (a.order[k] == 1) return k;
    return -1;
}

// fixed-order block sum (deterministic): wave DPP-free shuffle tree, then lane 0 of wave 0 adds the waves
__device__ __forceinline__ float block_sum(float v, float* lds) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int k = 0; k < JT_THREADS / 64; k++) s += lds[k];
    return s;
}

__global__ __launch_bounds__(JT_THREADS) void jitter_mean_kernel(JitterArgs a, int slot) {
    __shared__ float lds[JT_THREADS / 64];
    const int n = blockIdx.y;
    const float* base = a.img + (size_t)n * 3 * a.HW;
    const int per = (a.HW + JT_PARTS - 1) / JT_PARTS;
    const int p0 = blockIdx.x * per, p1 = min(p0 + per, a.HW);
    float acc = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += JT_THREADS) {
        float r = base[p], g = base[a.HW + p], b = base[2 * a.HW + p];
        chain(a, 0, slot, 0.f, r, g, b);
        acc += gray(r, g, b);
    }
    const float s = block_sum(acc, lds);
    if (threadIdx.x == 0) a.partial[n * JT_PARTS + blockIdx.x] = s;
}

__global__ __launch_bounds__(JT_THREADS) void jitter_apply_kernel(JitterArgs a, int slot) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * JT_THREADS + threadIdx.x;
    float cmean = 0.f;
    if (slot >= 0) {  // every thread folds the 64 partials in the same order -> one value per image
        const float* part = a.partial + n * JT_PARTS;
        float s = 0.f;
        for (int k = 0; k < JT_PARTS; k++) s += part[k];
        cmean = s / (float)a.HW;
    }
    if (p >= a.HW) return;
    const float* base = a.img + (size_t)n * 3 * a.HW;
    float r = base[p], g = base[a.HW + p], b = base[2 * a.HW + p];
    chain(a, 0, 4, cmean, r, g, b);
    r = (r - a.mean[0]) / a.stdv[0];
    g = (g - a.mean[1]) / a.stdv[1];
    b = (b - a.mean[2]) / a.stdv[2];
    if (a.nhwc) {
        float* o = a.out + ((size_t)n * a.HW + p) * 3;
        o[0] = r; o[1] = g; o[2] = b;
    } else {
        float* o = a.out + (size_t)n * 3 * a.HW + p;
        o[0] = r; o[a.HW] = g; o[2 * (size_t)a.HW] = b;
    }
}

}  // namespace

extern "C" size_t scp_color_jitter_workspace(int N) { return (size_t)N * JT_PARTS * sizeof(float); }

extern "C" int scp_color_jitter_normalize(const float* img, int N, int H, int W, const int* order,
                                          const float* ratio, const float* one_minus, float hue_shift,
                                          const float* mean, const float* stdv, int out_nhwc, float* out,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0) return scp::fail(hipErrorInvalidValue, "color_jitter: empty problem");
    if (!img || !out || !order || !ratio || !one_minus || !mean || !stdv)
        return scp::fail(hipErrorInvalidValue, "color_jitter: null argument");
    JitterArgs a;
    a.img = img; a.out = out; a.partial = static_cast<float*>(workspace);
    a.N = N; a.HW = H * W; a.nhwc = out_nhwc;
    int seen = 0;
    for (int k = 0; k < 4; k++) {
        a.order[k] = order[k];
        if (order[k] < -1 || order[k] > 3) return scp::fail(hipErrorInvalidValue, "color_jitter: op id out of range");
        if (order[k] >= 0) {
            if (seen & (1 << order[k])) return scp::fail(hipErrorInvalidValue, "color_jitter: op listed twice");
            seen |= 1 << order[k];
        }
    }
    for (int k = 0; k < 3; k++) {
        a.ratio[k] = ratio[k]; a.one_minus[k] = one_minus[k]; a.mean[k] = mean[k]; a.stdv[k] = stdv[k];
    }
    a.hue_shift = hue_shift;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int slot = -1;
    for (int k = 0; k < 4; k++)
        if (a.order[k] == 1) slot = k;
    if (slot >= 0) {
        if (!workspace || workspace_bytes < scp_color_jitter_workspace(N))
            return scp::fail(hipErrorInvalidValue, "color_jitter: workspace too small");
        hipLaunchKernelGGL(jitter_mean_kernel, dim3(JT_PARTS, N), dim3(JT_THREADS), 0, st, a, slot);
        if (int e = scp::check_launch("color_jitter mean")) return e;
    }
    hipLaunchKernelGGL(jitter_apply_kernel, dim3((a.HW + JT_THREADS - 1) / JT_THREADS, N), dim3(JT_THREADS), 0, st,
                       a, slot);
    return scp::check_launch("color_jitter apply");
}
