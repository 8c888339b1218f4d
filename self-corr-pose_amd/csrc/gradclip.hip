// self-corr-pose_amd/csrc/gradclip.hip -- per-group gradient clipping + NaN guard on the flat gradient buffer, two launches.
//
// Replaces model/trainer.py:132-150 (collect_grad: clip_grad_norm_ per parameter group -- mean_v 1.0, shapenerf 1.0, pose_predictor 0.1 --
// and "any non-finite gradient => zero_grad()") as the trainer runs it on scp_amd.parallel.FlatGradients' buffer.  As torch ops that was
// ~25 launches and 8 passes over the 58 MB buffer between backward and AdamW; here: one reduction pass (sum of squares per group +
// non-finite flag; per-block partials folded in block order by the last-arriving block: deterministic) and one apply pass
// (g <- finite ? prescale * coef[group(i)] * g : 0 with coef = min(1, max_norm / (norm + 1e-6)), clip_grad_norm_'s coefficient).
// `prescale` = 1 / world for the averaged all-reduce.  HBM-bound: 4 B read + (4 B read + 4 B write) per element.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {
constexpr int MAXR = SCP_GRADCLIP_MAX_RANGES;
struct Ranges {
    long long begin[MAXR], end[MAXR];
    int group[MAXR];
    int n;
};

__device__ __forceinline__ int group_of(const Ranges& r, long long i) {
    int g = -1;
#pragma unroll 4
    for (int k = 0; k < r.n; k++)
        if (i >= r.begin[k] && i < r.end[k]) g = r.group[k];
    return g;
}

// group of element i and, in `same_until`, the first index > i at which the group may change (the nearest range boundary above i)
__device__ __forceinline__ int group_span(const Ranges& r, long long i, long long& same_until) {
    int g = -1;
    long long nb = 0x7fffffffffffffffLL;
#pragma unroll 4
    for (int k = 0; k < r.n; k++) {
        const long long b = r.begin[k], e = r.end[k];
        if (i >= b && i < e) g = r.group[k];
        if (b > i && b < nb) nb = b;
        if (e > i && e < nb) nb = e;
    }
    same_until = nb;
    return g;
}

// The buffer as [head | 16-byte aligned float4 chunks | tail]: head / tail are at most 3 elements each, taken by the first lanes of block 0.
struct Split {
    long long head, chunks;     // elements before the first aligned chunk; number of float4 chunks
};
__device__ __forceinline__ Split split_of(const float* flat, long long n) {
    Split s;
    s.head = (long long)(((16 - (reinterpret_cast<size_t>(flat) & 15)) & 15) >> 2);
    if (s.head > n) s.head = n;
    s.chunks = (n - s.head) >> 2;
    return s;
}

constexpr int UNROLL = 4;       // float4 loads in flight per lane: the loop is latency-bound without them (one 4-byte load per lane and
                                // iteration took 234 us for 58 MB)

__device__ __forceinline__ void accumulate(const Ranges& r, long long i, float g, double (&ss)[3], int& bad) {
    if (!isfinite(g)) { bad = 1; return; }
    const int k = group_of(r, i);
    if (k >= 0) ss[k] += (double)g * (double)g;
}

__global__ __launch_bounds__(256) void gradclip_reduce_kernel(const float* __restrict__ flat, long long n, float prescale, Ranges r,
                                                              float max0, float max1, float max2, double* __restrict__ partial,
                                                              unsigned* __restrict__ ticket, float* __restrict__ result) {
    double ss[3] = {0.0, 0.0, 0.0};
    int bad = 0;
    const Split sp = split_of(flat, n);
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(flat + sp.head);
    const long long stride = (long long)gridDim.x * 256;
    for (long long c0 = (long long)blockIdx.x * 256 + threadIdx.x; c0 < sp.chunks; c0 += stride * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const long long c = c0 + u * stride;
            v[u] = c < sp.chunks ? v4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const long long c = c0 + u * stride;
            if (c >= sp.chunks) break;
            const long long i = sp.head + 4 * c;
            const float g0 = v[u].x * prescale, g1 = v[u].y * prescale, g2 = v[u].z * prescale, g3 = v[u].w * prescale;
            long long same_until;
            const int k = group_span(r, i, same_until);
            if (i + 3 < same_until) {            // one group for the four (nearly always: the ranges are whole parameter groups)
                if (!(isfinite(g0) && isfinite(g1) && isfinite(g2) && isfinite(g3))) {
                    bad = 1;                      // non-finite elements add nothing, the finite ones still count (as element by element)
                    if (k >= 0) {
                        if (isfinite(g0)) ss[k] += (double)g0 * (double)g0;
                        if (isfinite(g1)) ss[k] += (double)g1 * (double)g1;
                        if (isfinite(g2)) ss[k] += (double)g2 * (double)g2;
                        if (isfinite(g3)) ss[k] += (double)g3 * (double)g3;
                    }
                } else if (k >= 0) {
                    ss[k] += ((double)g0 * (double)g0 + (double)g1 * (double)g1) + ((double)g2 * (double)g2 + (double)g3 * (double)g3);
                }
            } else {
                accumulate(r, i, g0, ss, bad); accumulate(r, i + 1, g1, ss, bad);
                accumulate(r, i + 2, g2, ss, bad); accumulate(r, i + 3, g3, ss, bad);
            }
        }
    }
    if (blockIdx.x == 0) {                        // head and tail elements
        const long long tail0 = sp.head + 4 * sp.chunks;
        if ((long long)threadIdx.x < sp.head) accumulate(r, threadIdx.x, flat[threadIdx.x] * prescale, ss, bad);
        const long long t = tail0 + threadIdx.x;
        if (t < n) accumulate(r, t, flat[t] * prescale, ss, bad);
    }
    __shared__ double red[4][256];
    for (int k = 0; k < 3; k++) red[k][threadIdx.x] = ss[k];
    red[3][threadIdx.x] = (double)bad;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 4; k++) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    __shared__ bool last;
    if (threadIdx.x == 0) {
        for (int k = 0; k < 4; k++) __hip_atomic_store(&partial[(size_t)blockIdx.x * 4 + k], red[k][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // the last-arriving block folds the per-block partials: lane t takes blocks t, t + 256, ... in that order, then the same LDS tree --
    // a fixed order whatever block arrives last (one lane walking all 1024 x 4 values was most of the kernel's time)
    double tot[4] = {0.0, 0.0, 0.0, 0.0};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256)
        for (int k = 0; k < 4; k++) tot[k] += __hip_atomic_load(&partial[(size_t)b * 4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int k = 0; k < 4; k++) red[k][threadIdx.x] = tot[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 4; k++) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const bool finite = red[3][0] == 0.0;
        const float mx[3] = {max0, max1, max2};
        for (int k = 0; k < 3; k++) {
            const float norm = finite ? (float)sqrt(red[k][0]) : 0.f;
            result[k] = norm;                                                  // what collect_grad returns
            result[3 + k] = fminf(mx[k] / (norm + 1e-6f), 1.0f);                // clip_grad_norm_'s coefficient
        }
        result[6] = finite ? 1.f : 0.f;
        *ticket = 0u;                                                          // ready for the next step
    }
}

__device__ __forceinline__ float applied(const Ranges& r, long long i, float v, float prescale, const float (&c)[3]) {
    const int k = group_of(r, i);
    const float g = v * prescale;
    return k >= 0 ? g * c[k] : g;
}

__global__ __launch_bounds__(256) void gradclip_apply_kernel(float* __restrict__ flat, long long n, float prescale, Ranges r,
                                                             const float* __restrict__ result) {
    const float finite = result[6];
    const float c[3] = {result[3], result[4], result[5]};
    const Split sp = split_of(flat, n);
    float4* __restrict__ v4 = reinterpret_cast<float4*>(flat + sp.head);
    const long long stride = (long long)gridDim.x * 256;
    for (long long c0 = (long long)blockIdx.x * 256 + threadIdx.x; c0 < sp.chunks; c0 += stride * UNROLL) {
        if (finite == 0.f) {                      // a non-finite gradient anywhere: every gradient becomes zero (the reference's zero_grad())
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (c0 + u * stride < sp.chunks) v4[c0 + u * stride] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const long long cc = c0 + u * stride;
            v[u] = cc < sp.chunks ? v4[cc] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const long long cc = c0 + u * stride;
            if (cc >= sp.chunks) break;
            const long long i = sp.head + 4 * cc;
            long long same_until;
            const int k = group_span(r, i, same_until);
            float4 o;
            if (i + 3 < same_until) {
                const float f = k >= 0 ? c[k] : 1.f;
                // (v * prescale) * coef, the element-wise order; groups outside every range keep g = v * prescale
                o.x = k >= 0 ? (v[u].x * prescale) * f : v[u].x * prescale;
                o.y = k >= 0 ? (v[u].y * prescale) * f : v[u].y * prescale;
                o.z = k >= 0 ? (v[u].z * prescale) * f : v[u].z * prescale;
                o.w = k >= 0 ? (v[u].w * prescale) * f : v[u].w * prescale;
            } else {
                o.x = applied(r, i, v[u].x, prescale, c); o.y = applied(r, i + 1, v[u].y, prescale, c);
                o.z = applied(r, i + 2, v[u].z, prescale, c); o.w = applied(r, i + 3, v[u].w, prescale, c);
            }
            v4[cc] = o;
        }
    }
    if (blockIdx.x == 0) {
        const long long tail0 = sp.head + 4 * sp.chunks;
        if ((long long)threadIdx.x < sp.head) flat[threadIdx.x] = finite == 0.f ? 0.f : applied(r, threadIdx.x, flat[threadIdx.x], prescale, c);
        const long long t = tail0 + threadIdx.x;
        if (t < n) flat[t] = finite == 0.f ? 0.f : applied(r, t, flat[t], prescale, c);
    }
}
}  // namespace

extern "C" size_t scp_gradclip_workspace(void) { return (size_t)SCP_GRADCLIP_BLOCKS * 4 * sizeof(double) + 64; }

extern "C" int scp_gradclip(float* flat, long long n, float prescale, const long long* begin, const long long* end, const int* group,
                            int nranges, float max_norm0, float max_norm1, float max_norm2, void* workspace, size_t workspace_bytes,
                            float* result, void* stream) {
    if (nranges < 0 || nranges > MAXR) return scp::fail(hipErrorInvalidValue, "scp_gradclip: too many ranges");
    if (workspace_bytes < scp_gradclip_workspace()) return scp::fail(hipErrorInvalidValue, "scp_gradclip: workspace too small");
    Ranges r;
    r.n = nranges;
    for (int k = 0; k < nranges; k++) {
        if (group[k] < 0 || group[k] > 2 || begin[k] > end[k]) return scp::fail(hipErrorInvalidValue, "scp_gradclip: bad range");
        r.begin[k] = begin[k]; r.end[k] = end[k]; r.group[k] = group[k];
    }
    for (int k = nranges; k < MAXR; k++) { r.begin[k] = r.end[k] = 0; r.group[k] = 0; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partial = static_cast<double*>(workspace);
    unsigned* ticket = reinterpret_cast<unsigned*>(partial + (size_t)SCP_GRADCLIP_BLOCKS * 4);
    hipLaunchKernelGGL(gradclip_reduce_kernel, dim3(SCP_GRADCLIP_BLOCKS), dim3(256), 0, st, flat, n, prescale, r, max_norm0, max_norm1,
                       max_norm2, partial, ticket, result);
    hipLaunchKernelGGL(gradclip_apply_kernel, dim3(SCP_GRADCLIP_BLOCKS * 2), dim3(256), 0, st, flat, n, prescale, r, result);
    return scp::check_launch("gradclip");
}
