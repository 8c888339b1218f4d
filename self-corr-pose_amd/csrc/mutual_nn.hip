// self-corr-pose_amd/csrc/mutual_nn.hip -- row and column argmax of a masked score matrix in ONE pass.
//
// Replaces `pointcorr = pointcorr * (mask > 0) - 1e5 * (mask == 0); bw = pointcorr.max(1).indices;
// fw = pointcorr.max(2).indices` of PretrainedCorrespondence.match (model/module/pretrained_corr.py:85-89): the
// mutual nearest neighbours between the DINO key features of two images.  As torch ops this is six passes over the
// [N,P,Q] score tensor (outer-product mask, compare, select, two max reductions; 268 MB per pass at N=64, P=Q=1024).
// Here each workgroup owns a strip of 32 rows of one pair: every thread keeps the running (value, row) best of its 4
// columns in registers and the strip's row maxima are reduced across the workgroup; column partials of the strips are
// merged with a 64-bit atomicMax on (order-preserving float bits << 32 | ~row), row results are written directly.
// Ties resolve to the LOWEST index in both directions (the CPU semantics of torch.max).  HBM-bound: one read of S.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr int MN_THREADS = 256;
constexpr int MN_ROWS = 32;

__device__ __forceinline__ unsigned ordered(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(MN_THREADS) void mutual_argmax_kernel(const float* __restrict__ S,
                                                                   const float* __restrict__ rowmask,
                                                                   const float* __restrict__ colmask, int P, int Q,
                                                                   unsigned long long* __restrict__ colbest,
                                                                   long long* __restrict__ row_index) {
    __shared__ unsigned long long wave_best[MN_THREADS / 64];
    const int n = blockIdx.y;
    const int r0 = blockIdx.x * MN_ROWS, r1 = min(r0 + MN_ROWS, P);
    const float* Sn = S + (size_t)n * P * Q;
    for (int c0 = 0; c0 < Q; c0 += 4 * MN_THREADS) {            // column chunks of 1024 (one for the 256^2 config)
        const int c = c0 + 4 * threadIdx.x;
        const bool live = c < Q;
        float4 cm = {0, 0, 0, 0};
        if (live && colmask) cm = *reinterpret_cast<const float4*>(colmask + (size_t)n * Q + c);
        const bool k0 = !colmask || cm.x > 0, k1 = !colmask || cm.y > 0, k2 = !colmask || cm.z > 0, k3 = !colmask || cm.w > 0;
        float bv0 = -INFINITY, bv1 = -INFINITY, bv2 = -INFINITY, bv3 = -INFINITY;
        int br0 = 0, br1 = 0, br2 = 0, br3 = 0;
        for (int r = r0; r < r1; r++) {
            const bool rk = !rowmask || rowmask[(size_t)n * P + r] > 0;
            float4 v = {-1e5f, -1e5f, -1e5f, -1e5f};
            if (live) {
                const float4 s = *reinterpret_cast<const float4*>(Sn + (size_t)r * Q + c);
                if (rk && k0) v.x = s.x;
                if (rk && k1) v.y = s.y;
                if (rk && k2) v.z = s.z;
                if (rk && k3) v.w = s.w;
            }
            // columns: strictly greater keeps the earliest row
            if (v.x > bv0) { bv0 = v.x; br0 = r; }
            if (v.y > bv1) { bv1 = v.y; br1 = r; }
            if (v.z > bv2) { bv2 = v.z; br2 = r; }
            if (v.w > bv3) { bv3 = v.w; br3 = r; }
            // row: best of this thread's 4 columns (lowest column on ties), then across the workgroup
            float rv = v.x; int rc = c;
            if (v.y > rv) { rv = v.y; rc = c + 1; }
            if (v.z > rv) { rv = v.z; rc = c + 2; }
            if (v.w > rv) { rv = v.w; rc = c + 3; }
            unsigned long long key = live ? (((unsigned long long)ordered(rv) << 32) | (unsigned)(~(unsigned)rc)) : 0ull;
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(key, off, 64);
                key = o > key ? o : key;
            }
            if ((threadIdx.x & 63) == 0) wave_best[threadIdx.x >> 6] = key;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long b = wave_best[0];
                for (int w = 1; w < MN_THREADS / 64; w++) b = wave_best[w] > b ? wave_best[w] : b;
                // chunks of one row are visited in increasing column order: keep the earlier chunk on ties
                unsigned long long* slot = reinterpret_cast<unsigned long long*>(row_index) + (size_t)n * P + r;
                if (c0 == 0 || b > *slot) *slot = b;
            }
            __syncthreads();
        }
        if (live) {
            unsigned long long* cb = colbest + (size_t)n * Q + c;
            atomicMax(cb + 0, ((unsigned long long)ordered(bv0) << 32) | (unsigned)(~(unsigned)br0));
            atomicMax(cb + 1, ((unsigned long long)ordered(bv1) << 32) | (unsigned)(~(unsigned)br1));
            atomicMax(cb + 2, ((unsigned long long)ordered(bv2) << 32) | (unsigned)(~(unsigned)br2));
            atomicMax(cb + 3, ((unsigned long long)ordered(bv3) << 32) | (unsigned)(~(unsigned)br3));
        }
    }
}

// packed keys -> indices (in place for the row slots, which were written as keys; out of place for the columns)
__global__ void mutual_unpack_kernel(const unsigned long long* __restrict__ colbest, long total_cols,
                                     long long* __restrict__ col_index, long long* __restrict__ row_index, long total_rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total_cols) col_index[i] = (long long)(unsigned)(~(unsigned)(colbest[i] & 0xFFFFFFFFull));
    if (i < total_rows) {
        const unsigned long long k = reinterpret_cast<unsigned long long*>(row_index)[i];
        row_index[i] = (long long)(unsigned)(~(unsigned)(k & 0xFFFFFFFFull));
    }
}

}  // namespace

extern "C" size_t scp_mutual_argmax_workspace(int N, int Q) { return (size_t)N * Q * sizeof(unsigned long long); }

extern "C" int scp_mutual_argmax(const float* scores, const float* rowmask, const float* colmask, int N, int P, int Q,
                                 long long* col_index, long long* row_index, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    if (N <= 0 || P <= 0 || Q <= 0) return scp::fail(hipErrorInvalidValue, "mutual_argmax: empty problem");
    if (Q % 4 != 0) return scp::fail(hipErrorInvalidValue, "mutual_argmax: Q must be a multiple of 4");
    if (!scores || !col_index || !row_index) return scp::fail(hipErrorInvalidValue, "mutual_argmax: null argument");
    if (!workspace || workspace_bytes < scp_mutual_argmax_workspace(N, Q))
        return scp::fail(hipErrorInvalidValue, "mutual_argmax: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long* colbest = static_cast<unsigned long long*>(workspace);
    if (hipMemsetAsync(colbest, 0, (size_t)N * Q * sizeof(unsigned long long), st) != hipSuccess)
        return scp::check_launch("mutual_argmax memset");
    hipLaunchKernelGGL(mutual_argmax_kernel, dim3((P + MN_ROWS - 1) / MN_ROWS, N), dim3(MN_THREADS), 0, st, scores, rowmask,
                       colmask, P, Q, colbest, row_index);
    if (int e = scp::check_launch("mutual_argmax")) return e;
    const long tc = (long)N * Q, tr = (long)N * P;
    const long total = tc > tr ? tc : tr;
    hipLaunchKernelGGL(mutual_unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, colbest, tc, col_index,
                       row_index, tr);
    return scp::check_launch("mutual_argmax unpack");
}
