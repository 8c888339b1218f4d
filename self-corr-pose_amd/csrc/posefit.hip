// self-corr-pose_amd/csrc/posefit.hip -- test-time pose fitting: batched RANSAC + Umeyama similarity fit.
//
// Replaces model/util/umeyama.py:9-38,97-131,177-201 as driven by Tester.pose_fitting
// (model/tester.py:346-384): per image, 100 RANSAC rounds of {5-point Umeyama fit, residual of ALL n
// correspondences, inlier set}, then one Umeyama fit on the best round's inliers.  The reference runs this
// as ~10 tiny torch launches + 2 host syncs per round per image (3200 rounds for a batch of 32).  Here the
// whole batch is three launches:
//   hypotheses : one thread per (image, round): gather 5 pairs -> moments -> 3x3 SVD (one-sided Jacobi, fp64)
//                -> similarity transform
//   score      : one THREAD per round, one workgroup per (point tile, image); the tile's points are staged in
//                LDS and broadcast to all rounds, so each thread owns its round's residual sum and inlier
//                count and no cross-lane reduction is needed; tile partials are folded in a fixed order
//   fit        : inlier moments of the chosen round (22 fp64 sums, block-reduced per tile), then the same
//                moments -> transform routine
// Point arithmetic is fp32 like the reference's (`TargetHom - OutTransform @ SourceHom`, norm, `<`), sums are
// fp64.  The 3x3 SVD runs in fp64 so that R = (U Vh)^T, sum(D) agree with LAPACK's fp32 result to rounding
// wherever the 5-point configuration is not degenerate.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr int PF_TILE = 2048;       // points per workgroup in score / fit
constexpr int PF_FIT_THREADS = 256;

struct Moments {
    double n;
    double s[3], t[3];   // sum source, sum target
    double ss[3];        // sum source^2 per axis
    double ts[9];        // sum target_i * source_j
};

struct Similarity {
    double scale;
    double rot[9];       // "Rotation" of the reference = (U Vh)^T, row-major
    double trans[3];
};

__device__ inline void rotate_cols(double a[9], double v[9], int p, int q) {
    double alpha = 0, beta = 0, gamma = 0;
    for (int i = 0; i < 3; i++) {
        alpha += a[3 * i + p] * a[3 * i + p];
        beta += a[3 * i + q] * a[3 * i + q];
        gamma += a[3 * i + p] * a[3 * i + q];
    }
    if (gamma == 0.0 || fabs(gamma) <= 1e-300) return;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
    for (int i = 0; i < 3; i++) {
        const double ap = a[3 * i + p], aq = a[3 * i + q];
        a[3 * i + p] = c * ap - s * aq;
        a[3 * i + q] = s * ap + c * aq;
        const double vp = v[3 * i + p], vq = v[3 * i + q];
        v[3 * i + p] = c * vp - s * vq;
        v[3 * i + q] = s * vp + c * vq;
    }
}

// A = U diag(d) V^T, d descending, U and V orthogonal (columns completed by cross products when d ~ 0)
__device__ inline void svd3(const double A[9], double U[9], double d[3], double V[9]) {
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++) a[i] = A[i];
    for (int sweep = 0; sweep < 30; sweep++) {
        rotate_cols(a, v, 0, 1);
        rotate_cols(a, v, 0, 2);
        rotate_cols(a, v, 1, 2);
        double off = 0, diag = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 3; q++) {
                double g = 0;
                for (int i = 0; i < 3; i++) g += a[3 * i + p] * a[3 * i + q];
                off += g * g;
            }
        for (int p = 0; p < 3; p++) {
            double g = 0;
            for (int i = 0; i < 3; i++) g += a[3 * i + p] * a[3 * i + p];
            diag += g * g;
        }
        if (off <= 1e-30 * diag) break;
    }
    double nrm[3];
    int ord[3] = {0, 1, 2};
    for (int p = 0; p < 3; p++) nrm[p] = sqrt(a[p] * a[p] + a[3 + p] * a[3 + p] + a[6 + p] * a[6 + p]);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2 - i; j++)
            if (nrm[ord[j]] < nrm[ord[j + 1]]) { const int t = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = t; }
    const double tiny = 1e-12 * (nrm[ord[0]] > 0 ? nrm[ord[0]] : 1.0);
    for (int k = 0; k < 3; k++) {
        const int p = ord[k];
        d[k] = nrm[p];
        for (int i = 0; i < 3; i++) {
            V[3 * i + k] = v[3 * i + p];
            U[3 * i + k] = nrm[p] > tiny ? a[3 * i + p] / nrm[p] : 0.0;
        }
    }
    // complete U when trailing singular values vanish (rank-deficient covariance of 5 near-coplanar points)
    if (!(d[0] > tiny)) { U[0] = 1; U[3] = 0; U[6] = 0; }
    if (!(d[1] > tiny)) {
        // any unit vector orthogonal to column 0
        const double x = U[0], y = U[3], z = U[6];
        double e[3] = {0, 0, 0};
        if (fabs(x) <= fabs(y) && fabs(x) <= fabs(z)) e[0] = 1; else if (fabs(y) <= fabs(z)) e[1] = 1; else e[2] = 1;
        double c[3] = {y * e[2] - z * e[1], z * e[0] - x * e[2], x * e[1] - y * e[0]};
        const double n = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        U[1] = c[0] / n; U[4] = c[1] / n; U[7] = c[2] / n;
    }
    if (!(d[2] > tiny)) {
        U[2] = U[3] * U[7] - U[6] * U[4];
        U[5] = U[6] * U[1] - U[0] * U[7];
        U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

__device__ inline double det3(const double m[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// estimateSimilarityUmeyama (umeyama.py:161-201) from raw moments
__device__ inline Similarity umeyama(const Moments& m) {
    Similarity out;
    const double n = m.n;
    double sb[3], tb[3];
    for (int i = 0; i < 3; i++) { sb[i] = m.s[i] / n; tb[i] = m.t[i] / n; }
    double cov[9];  // CenteredTarget @ CenteredSource^T / nPoints  (:172)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) cov[3 * i + j] = (m.ts[3 * i + j] - n * tb[i] * sb[j]) / n;
    double U[9], D[3], V[9];
    svd3(cov, U, D, V);
    double Vh[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Vh[3 * i + j] = V[3 * j + i];
    if (det3(U) * det3(Vh) < 0.0) {  // :182-185
        D[2] = -D[2];
        U[2] = -U[2]; U[5] = -U[5]; U[8] = -U[8];
    }
    double uv[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) uv[3 * i + j] = U[3 * i] * Vh[j] + U[3 * i + 1] * Vh[3 + j] + U[3 * i + 2] * Vh[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out.rot[3 * i + j] = uv[3 * j + i];  // .T (:187)
    double varp = 0;  // torch.var (unbiased) per axis, summed (:189)
    for (int i = 0; i < 3; i++) varp += (m.ss[i] - n * sb[i] * sb[i]) / (n - 1.0);
    out.scale = (D[0] + D[1] + D[2]) / varp;  // :190
    // Translation = mean(target) - mean(source)[None].mm(ScaleFact * Rotation)  (:193): row vector times matrix
    for (int j = 0; j < 3; j++)
        out.trans[j] = tb[j] - out.scale * (sb[0] * out.rot[j] + sb[1] * out.rot[3 + j] + sb[2] * out.rot[6 + j]);
    return out;
}

// OutTransform rows 0..2 (row-major 3x4): [ScaleFact * Rotation | Translation]  (:195-197)
__device__ inline void store_transform(const Similarity& s, float* t12) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) t12[4 * i + j] = (float)(s.scale * s.rot[3 * i + j]);
        t12[4 * i + 3] = (float)s.trans[i];
    }
}

__global__ void hypotheses_kernel(const float* __restrict__ src, const float* __restrict__ tgt, int B, int Nmax,
                                  const int* __restrict__ rand_idx, int K, float* __restrict__ transforms) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * K) return;
    const int b = id / K;
    Moments m = {};
    m.n = 5.0;
    for (int p = 0; p < 5; p++) {
        const int j = rand_idx[(size_t)id * 5 + p];
        const float* s = src + ((size_t)b * Nmax + j) * 3;
        const float* t = tgt + ((size_t)b * Nmax + j) * 3;
        for (int i = 0; i < 3; i++) {
            m.s[i] += (double)s[i];
            m.t[i] += (double)t[i];
            m.ss[i] += (double)s[i] * (double)s[i];
            for (int k = 0; k < 3; k++) m.ts[3 * i + k] += (double)t[i] * (double)s[k];
        }
    }
    store_transform(umeyama(m), transforms + (size_t)id * 12);
}

// residual of one correspondence under a 3x4 transform, fp32 like `Diff = TargetHom - OutTransform @ SourceHom`
__device__ __forceinline__ float residual(const float* T, float sx, float sy, float sz, float tx, float ty, float tz) {
    const float dx = tx - (T[0] * sx + T[1] * sy + T[2] * sz + T[3]);
    const float dy = ty - (T[4] * sx + T[5] * sy + T[6] * sz + T[7]);
    const float dz = tz - (T[8] * sx + T[9] * sy + T[10] * sz + T[11]);
    return sqrtf(dx * dx + dy * dy + dz * dz);
}

__global__ void score_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                             const int* __restrict__ counts, int Nmax, const float* __restrict__ transforms, int K,
                             const float* __restrict__ pass_thr, int tiles, double* __restrict__ part_sq,
                             int* __restrict__ part_in) {
    __shared__ float pts[PF_TILE * 6];
    const int b = blockIdx.y, tile = blockIdx.x;
    const int n = counts[b];
    const int p0 = tile * PF_TILE;
    const int cnt = max(0, min(PF_TILE, n - p0));
    for (int i = threadIdx.x; i < cnt * 3; i += blockDim.x) {
        const int p = i / 3, c = i - 3 * p;
        pts[6 * p + c] = src[((size_t)b * Nmax + p0) * 3 + i];
        pts[6 * p + 3 + c] = tgt[((size_t)b * Nmax + p0) * 3 + i];
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k >= K) return;
    float T[12];
    for (int i = 0; i < 12; i++) T[i] = transforms[((size_t)b * K + k) * 12 + i];
    const float thr = pass_thr[b];
    double sq = 0.0;
    int inl = 0;
    for (int p = 0; p < cnt; p++) {
        const float r = residual(T, pts[6 * p], pts[6 * p + 1], pts[6 * p + 2], pts[6 * p + 3], pts[6 * p + 4], pts[6 * p + 5]);
        sq += (double)r * (double)r;
        inl += r < thr;
    }
    part_sq[((size_t)b * tiles + tile) * K + k] = sq;
    part_in[((size_t)b * tiles + tile) * K + k] = inl;
}

__global__ void score_fold_kernel(const double* __restrict__ part_sq, const int* __restrict__ part_in, int B, int K,
                                  int tiles, double* __restrict__ residual_sq, int* __restrict__ inliers) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * K) return;
    const int b = id / K, k = id - b * K;
    double sq = 0.0;
    int inl = 0;
    for (int t = 0; t < tiles; t++) {
        sq += part_sq[((size_t)b * tiles + t) * K + k];
        inl += part_in[((size_t)b * tiles + t) * K + k];
    }
    residual_sq[id] = sq;
    inliers[id] = inl;
}

constexpr int PF_NM = 22;  // doubles in a Moments record

__global__ __launch_bounds__(PF_FIT_THREADS) void fit_moments_kernel(
    const float* __restrict__ src, const float* __restrict__ tgt, const int* __restrict__ counts, int Nmax,
    const float* __restrict__ chosen, const float* __restrict__ pass_thr, int tiles, double* __restrict__ part) {
    __shared__ double lds[PF_FIT_THREADS / 64][PF_NM];
    const int b = blockIdx.y, tile = blockIdx.x;
    const int n = counts[b];
    const int p0 = tile * PF_TILE;
    const int p1 = min(p0 + PF_TILE, n);
    float T[12];
    for (int i = 0; i < 12; i++) T[i] = chosen[(size_t)b * 12 + i];
    const float thr = pass_thr[b];
    double m[PF_NM];
    for (int i = 0; i < PF_NM; i++) m[i] = 0.0;
    for (int p = p0 + threadIdx.x; p < p1; p += PF_FIT_THREADS) {
        const float* s = src + ((size_t)b * Nmax + p) * 3;
        const float* t = tgt + ((size_t)b * Nmax + p) * 3;
        if (residual(T, s[0], s[1], s[2], t[0], t[1], t[2]) < thr) {
            m[0] += 1.0;
            for (int i = 0; i < 3; i++) {
                m[1 + i] += (double)s[i];
                m[4 + i] += (double)t[i];
                m[7 + i] += (double)s[i] * (double)s[i];
                for (int k = 0; k < 3; k++) m[10 + 3 * i + k] += (double)t[i] * (double)s[k];
            }
        }
    }
    for (int i = 0; i < PF_NM; i++) {
        double v = m[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < PF_NM) {
        double v = 0.0;
        for (int w = 0; w < PF_FIT_THREADS / 64; w++) v += lds[w][threadIdx.x];
        part[((size_t)b * tiles + tile) * PF_NM + threadIdx.x] = v;
    }
}

__global__ void fit_solve_kernel(const double* __restrict__ part, int B, int tiles, float* __restrict__ scale,
                                 float* __restrict__ rotation, float* __restrict__ translation,
                                 float* __restrict__ transform, int* __restrict__ n_inliers) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double v[PF_NM];
    for (int i = 0; i < PF_NM; i++) v[i] = 0.0;
    for (int t = 0; t < tiles; t++)
        for (int i = 0; i < PF_NM; i++) v[i] += part[((size_t)b * tiles + t) * PF_NM + i];
    Moments m;
    m.n = v[0];
    for (int i = 0; i < 3; i++) { m.s[i] = v[1 + i]; m.t[i] = v[4 + i]; m.ss[i] = v[7 + i]; }
    for (int i = 0; i < 9; i++) m.ts[i] = v[10 + i];
    n_inliers[b] = (int)v[0];
    if (v[0] < 2.0) {  // nothing to fit: flag with NaNs, the host raises like the reference's failure path
        const float nan = __int_as_float(0x7fc00000);
        scale[b] = nan;
        for (int i = 0; i < 9; i++) rotation[b * 9 + i] = nan;
        for (int i = 0; i < 3; i++) translation[b * 3 + i] = nan;
        for (int i = 0; i < 16; i++) transform[b * 16 + i] = nan;
        return;
    }
    const Similarity s = umeyama(m);
    scale[b] = (float)s.scale;
    for (int i = 0; i < 9; i++) rotation[b * 9 + i] = (float)s.rot[i];
    for (int i = 0; i < 3; i++) translation[b * 3 + i] = (float)s.trans[i];
    float t12[12];
    store_transform(s, t12);
    for (int i = 0; i < 12; i++) transform[b * 16 + i] = t12[i];
    transform[b * 16 + 12] = 0.f; transform[b * 16 + 13] = 0.f; transform[b * 16 + 14] = 0.f; transform[b * 16 + 15] = 1.f;
}

int tiles_for(int Nmax) { return (Nmax + PF_TILE - 1) / PF_TILE; }

}  // namespace

extern "C" size_t scp_posefit_workspace(int B, int Nmax, int K) {
    if (B <= 0 || Nmax <= 0 || K <= 0) return 0;
    const size_t t = (size_t)tiles_for(Nmax);
    const size_t score = (size_t)B * t * K * (sizeof(double) + sizeof(int));
    const size_t fit = (size_t)B * t * PF_NM * sizeof(double);
    return (score > fit ? score : fit) + 64;
}

extern "C" int scp_ransac_hypotheses(const float* source, const float* target, int B, int Nmax, const int* rand_idx,
                                     int K, float* transforms, void* stream) {
    if (B <= 0 || Nmax <= 0 || K <= 0) return scp::fail(hipErrorInvalidValue, "ransac_hypotheses: empty problem");
    if (!source || !target || !rand_idx || !transforms) return scp::fail(hipErrorInvalidValue, "ransac_hypotheses: null argument");
    hipLaunchKernelGGL(hypotheses_kernel, dim3((B * K + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), source,
                       target, B, Nmax, rand_idx, K, transforms);
    return scp::check_launch("ransac_hypotheses");
}

extern "C" int scp_ransac_score(const float* source, const float* target, const int* counts, int B, int Nmax,
                                const float* transforms, int K, const float* pass_threshold, double* residual_sq,
                                int* inliers, void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0 || Nmax <= 0 || K <= 0) return scp::fail(hipErrorInvalidValue, "ransac_score: empty problem");
    if (K > 1024) return scp::fail(hipErrorInvalidValue, "ransac_score: at most 1024 hypotheses per problem");
    if (!source || !target || !counts || !transforms || !pass_threshold || !residual_sq || !inliers)
        return scp::fail(hipErrorInvalidValue, "ransac_score: null argument");
    if (workspace_bytes < scp_posefit_workspace(B, Nmax, K)) return scp::fail(hipErrorInvalidValue, "ransac_score: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles = tiles_for(Nmax);
    double* part_sq = static_cast<double*>(workspace);
    int* part_in = reinterpret_cast<int*>(part_sq + (size_t)B * tiles * K);
    const int threads = (K + 63) / 64 * 64;
    hipLaunchKernelGGL(score_kernel, dim3(tiles, B), dim3(threads), 0, st, source, target, counts, Nmax, transforms, K,
                       pass_threshold, tiles, part_sq, part_in);
    if (int e = scp::check_launch("ransac_score")) return e;
    hipLaunchKernelGGL(score_fold_kernel, dim3((B * K + 255) / 256), dim3(256), 0, st, part_sq, part_in, B, K, tiles,
                       residual_sq, inliers);
    return scp::check_launch("ransac_score fold");
}

extern "C" int scp_umeyama_fit_inliers(const float* source, const float* target, const int* counts, int B, int Nmax,
                                       const float* chosen, const float* pass_threshold, float* scale, float* rotation,
                                       float* translation, float* transform, int* n_inliers, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (B <= 0 || Nmax <= 0) return scp::fail(hipErrorInvalidValue, "umeyama_fit: empty problem");
    if (!source || !target || !counts || !chosen || !pass_threshold || !scale || !rotation || !translation || !transform ||
        !n_inliers)
        return scp::fail(hipErrorInvalidValue, "umeyama_fit: null argument");
    if (workspace_bytes < scp_posefit_workspace(B, Nmax, 1)) return scp::fail(hipErrorInvalidValue, "umeyama_fit: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles = tiles_for(Nmax);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(fit_moments_kernel, dim3(tiles, B), dim3(PF_FIT_THREADS), 0, st, source, target, counts, Nmax, chosen,
                       pass_threshold, tiles, part);
    if (int e = scp::check_launch("umeyama_fit moments")) return e;
    hipLaunchKernelGGL(fit_solve_kernel, dim3((B + 63) / 64), dim3(64), 0, st, part, B, tiles, scale, rotation, translation,
                       transform, n_inliers);
    return scp::check_launch("umeyama_fit solve");
}
