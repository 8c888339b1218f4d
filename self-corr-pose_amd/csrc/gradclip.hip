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

__global__ __launch_bounds__(256) void gradclip_reduce_kernel(const float* __restrict__ flat, long long n, float prescale, Ranges r,
                                                              float max0, float max1, float max2, double* __restrict__ partial,
                                                              unsigned* __restrict__ ticket, float* __restrict__ result) {
    double ss[3] = {0.0, 0.0, 0.0};
    int bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float g = flat[i] * prescale;
        if (!isfinite(g)) { bad = 1; continue; }
        const int k = group_of(r, i);
        if (k >= 0) ss[k] += (double)g * (double)g;
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
    if (threadIdx.x == 0) {
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
        for (unsigned b = 0; b < gridDim.x; b++)
            for (int k = 0; k < 4; k++) tot[k] += __hip_atomic_load(&partial[(size_t)b * 4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool finite = tot[3] == 0.0;
        const float mx[3] = {max0, max1, max2};
        for (int k = 0; k < 3; k++) {
            const float norm = finite ? (float)sqrt(tot[k]) : 0.f;
            result[k] = norm;                                                  // what collect_grad returns
            result[3 + k] = fminf(mx[k] / (norm + 1e-6f), 1.0f);                // clip_grad_norm_'s coefficient
        }
        result[6] = finite ? 1.f : 0.f;
        *ticket = 0u;                                                          // ready for the next step
    }
}

__global__ __launch_bounds__(256) void gradclip_apply_kernel(float* __restrict__ flat, long long n, float prescale, Ranges r,
                                                             const float* __restrict__ result) {
    const float finite = result[6];
    const float c[3] = {result[3], result[4], result[5]};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (finite == 0.f) { flat[i] = 0.f; continue; }
        const int k = group_of(r, i);
        const float g = flat[i] * prescale;
        flat[i] = k >= 0 ? g * c[k] : g;
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
