// self-corr-pose_amd/csrc/nearest.hip -- brute-force 1-nearest-neighbour (squared L2, 3-D points).
//
// Replaces pytorch3d.ops.knn_points(K=1) as used by the symmetry loss
// (model/module/mesh.py:53-62 -> model/util/chamfer.py:135: every vertex against 10 000 sampled
// surface points, k*B point clouds).  pytorch3d is an un-vendored CUDA dependency with no ROCm build.
// One thread per query point, the candidate cloud streams through LDS in 1024-point tiles (all lanes
// read the same candidate -> LDS broadcast); the candidate range is split over blockIdx.y so that
// >= 1k workgroups are in flight, partial (distance, index) results are merged with a 64-bit atomicMin
// on (float bits << 32 | index) -- distances are non-negative so the float bit pattern orders like
// the value and ties resolve to the lowest index, like a sequential argmin.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(NN_THREADS) void nearest_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ y, int P1, int P2,
                                                             int y_per_block,
                                                             unsigned long long* __restrict__ best) {
    __shared__ float ys[NN_TILE * 3];
    const int n = blockIdx.z;
    const int i = blockIdx.x * NN_THREADS + threadIdx.x;
    const bool ok = i < P1;
    const float* xp = x + ((size_t)n * P1 + (ok ? i : 0)) * 3;
    const float x0 = xp[0], x1 = xp[1], x2 = xp[2];
    const int j0 = blockIdx.y * y_per_block, j1 = min(j0 + y_per_block, P2);
    float dmin = INFINITY;
    int jmin = 0;
    for (int t = j0; t < j1; t += NN_TILE) {
        const int cnt = min(NN_TILE, j1 - t);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * 3; k += NN_THREADS) ys[k] = y[((size_t)n * P2 + t) * 3 + k];
        __syncthreads();
        for (int k = 0; k < cnt; k++) {
            const float a = x0 - ys[3 * k], b = x1 - ys[3 * k + 1], c = x2 - ys[3 * k + 2];
            const float d = a * a + b * b + c * c;
            if (d < dmin) { dmin = d; jmin = t + k; }
        }
    }
    if (ok) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(dmin) << 32) | (unsigned)jmin;
        atomicMin(best + (size_t)n * P1 + i, key);
    }
}

__global__ void nearest_unpack_kernel(const unsigned long long* __restrict__ best, long total,
                                      long long* __restrict__ idx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) idx[i] = (long long)(best[i] & 0xFFFFFFFFull);
}

}  // namespace

extern "C" size_t scp_nearest_point_workspace(int N, int P1) { return (size_t)N * P1 * sizeof(unsigned long long); }

extern "C" int scp_nearest_point(const float* x, const float* y, int N, int P1, int P2, long long* index,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (N <= 0 || P1 <= 0 || P2 <= 0) return scp::fail(hipErrorInvalidValue, "nearest_point: empty problem");
    if (workspace_bytes < scp_nearest_point_workspace(N, P1))
        return scp::fail(hipErrorInvalidValue, "nearest_point: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long* best = static_cast<unsigned long long*>(workspace);
    if (hipMemsetAsync(best, 0xFF, (size_t)N * P1 * sizeof(unsigned long long), st) != hipSuccess)
        return scp::check_launch("nearest_point memset");
    const int xblocks = (P1 + NN_THREADS - 1) / NN_THREADS;
    int ysplit = max(1, min((P2 + NN_TILE - 1) / NN_TILE, 2048 / max(1, xblocks * N)));
    const int y_per_block = ((P2 + ysplit - 1) / ysplit + NN_TILE - 1) / NN_TILE * NN_TILE;
    ysplit = (P2 + y_per_block - 1) / y_per_block;
    hipLaunchKernelGGL(nearest_kernel, dim3(xblocks, ysplit, N), dim3(NN_THREADS), 0, st, x, y, P1, P2,
                       y_per_block, best);
    if (int e = scp::check_launch("nearest_point")) return e;
    const long total = (long)N * P1;
    hipLaunchKernelGGL(nearest_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, best, total, index);
    return scp::check_launch("nearest_unpack");
}
