// self-corr-pose_amd/csrc/softras.hip -- SoftRas soft rasteriser for MI355X (gfx950), forward + backward.
//
// What it computes is fixed by the reference kernels
//   third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
//     :245-305 per-face precompute, :308-483 per-pixel forward, :486-668 per-pixel backward
// (semantics checklist: SURVEY.md Appendix A).  How it computes it is CDNA4-first:
//
//   * one 256-thread workgroup (4 wavefronts of 64) owns a pixel tile; the tile first bins the faces: 256 faces per
//     round are tested against the tile rectangle and compacted IN FACE-INDEX ORDER (ballot + popcount prefix) into an
//     LDS list -- order matters, the alpha product / online softmax / z-buffer tie-break are order dependent and must
//     equal the reference's brute-force loop bit for bit;
//   * listed faces are staged 32 at a time into LDS as 72-float records (corners, inverse, Gram matrix, obtuse flags,
//     vertex colours, and the face-only terms hoisted out of the pair loop: edge vectors of the Gram matrix, edge
//     denominators and the correctly rounded reciprocals of every face-constant divisor); nothing is re-read from L2
//     per pixel (the reference re-reads every face from global memory in every thread);
//   * forward: each wavefront owns an 8x8 quadrant; a face whose dilated bounding box misses the quadrant costs a single
//     LDS broadcast read + compare;
//   * backward works per (pixel, face) PAIR, not per face x 64 lanes: every wavefront scans the staged faces against
//     its pixels, appends the bbox-surviving pairs to a small LDS ring (each face's run padded to a multiple of 16
//     lanes) and runs the pair arithmetic on full groups of 64 queued pairs; per-pixel inputs come from an LDS pixel
//     table, per-face records are read with a per-row address.  The 18 partial derivatives are combined with a DPP
//     butterfly reduce-scatter inside every 16-lane row (one face per row by construction), added to an LDS accumulator
//     with ds_add_f32, and flushed with ONE global atomic per (tile, face, component) -- the reference issues 9-18
//     global atomics per (pixel, face);
//   * divisions by face / pass constants are exact without the IEEE division sequence: with y = RN(1/b) hoisted,
//     q = a*y followed by two fused residual corrections is the correctly rounded a/b (Markstein), 5 instructions
//     instead of ~11 and bit-identical to a/b (scp_selftest_exact_division checks it on the device);
//   * blockIdx is remapped so that the tiles of one image run on one XCD (its faces stay in that XCD's L2).
//
// Numerics: this file is compiled with -ffp-contract=off and without fast-math; every expression keeps the
// reference's evaluation order and its fp64 promotions where they change a value, so the only source
// of difference from the CPU oracle is the last-ulp behaviour of expf.  Explicit __builtin_fmaf calls are the exact-
// division residuals, never a contraction of the reference's arithmetic.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "scp_hip.h"
#include "scp_common.h"

namespace {

constexpr int TILE = 16;        // pixels per tile edge
constexpr int THREADS = 256;    // 4 wavefronts
constexpr int NB = 32;          // faces staged per round
constexpr int REC = 72;         // floats per staged face record (multiple of 4, and 72 mod 64 = 8: the four rows of a wavefront
                                // read four different records without sharing a bank)
constexpr int LIST_CAP = 1024;  // binned face ids held before a flush
constexpr int PIXREC = 20;      // floats per pixel-table entry (backward); 20*p mod 64 hits 16 distinct 4-bank groups
constexpr int QCAP = 256;       // pair-queue ring entries per wavefront (max outstanding 63 + 64 + 15)

// record layout (floats)
constexpr int R_V = 0;      // 9 corner coordinates
constexpr int R_INV = 9;    // 9
constexpr int R_SYM = 18;   // 9 (copied with R_INV / R_OBT in one run; read through R_EA)
[[maybe_unused]] constexpr int R_SYM_USED = R_SYM;
constexpr int R_OBT = 27;   // 3
constexpr int R_TEX = 30;   // 9 (vertex colours) -- only valid when texture_size == 3
constexpr int R_IDX = 39;   // face index (int bits)
constexpr int R_EA = 40;    // 9: edge e = (v0 = e, v1 = e+1 mod 3): a_j = sym[v0][j] - sym[v1][j]   (kernel.cu:41-44)
constexpr int R_EDEN = 49;  // 3: a[v0] - a[v1]  (the divisor of kernel.cu:45)
constexpr int R_ERCP = 52;  // 3: RN(1 / R_EDEN)
constexpr int R_ZRCP = 55;  // 3: RN(1 / z_corner)
constexpr int R_SLOW = 58;  // int bits: 1 = some face-constant divisor is zero / non-finite / extreme -> IEEE division
constexpr int R_PRE = 59;   // 3: conservative early-out thresholds m_k: a pixel with w_k < -m_k lies further than the dilation
                            //    margin behind edge line k and cannot pass the reference's `dis < threshold` test; +inf = off
constexpr int R_TEX2 = 62;  // 9: vertex colours of the fused hard-colour output (dual forward only)

// pixel-table layout (floats; backward)
constexpr int P_IMG = 0, P_GIMG = 4, P_SUM = 8, P_MAX = 9, P_XP = 10, P_YP = 11, P_RSUM = 12, P_SLOW = 13;

struct RasterArgs {
    const float* faces;
    const float* textures;
    const float* faces_info;
    float* aggrs_info;
    float* soft_colors;
    const float* textures2;      // dual forward: vertex colours, z-buffer info and image of the fused hard-colour pass
    float* aggrs_info2;
    float* soft_colors2;
    const float* soft_colors_in;
    const float* aggrs_info_in;
    const float* grad_soft_colors;
    float* grad_faces;
    float* grad_textures;
    unsigned long long* counter;
    int B, F, S, T, R;
    int tiles_per_row, tiles_per_image, total_tiles, tiles_per_xcd;
    float near_, far_, eps, sigma, dist_eps, gamma;
    float threshold, margin;
    float range, nrange;                                  // far - near, near - far (fp32, as the reference forms them)
    float rcp_sigma, rcp_gamma, rcp_range, rcp_nrange;    // RN(1 / x) of the pass constants
    int const_slow;                                       // a pass constant is outside the exact-division range
    int dist_mode, alpha_mode, double_side;
};

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float pick3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

// ------------------------------------------------------------------------------------------------
// exact division by a hoisted divisor.  y = RN(1/b) (one IEEE division per face / pixel / pass).  q0 = RN(a*y) is within
// 2 ulp of a/b; q1 = RN(q0 + r0*y) with the exact residual r0 = a - b*q0 is faithful; one more residual step gives the
// correctly rounded quotient (Markstein 1990, the same iteration the hardware's v_div_* sequence runs, minus its scaling
// and fix-up instructions).  Valid while no intermediate leaves the normal range: divisors are range-checked when they
// are hoisted; a pair whose face / pixel / pass constants fail the check runs the FAST = false instantiation of the pair
// arithmetic (plain a / b everywhere).  Verified against a / b on the device by scp_selftest_exact_division
// (tests/test_softras_gpu.py).
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ bool fast_div_range(float b) {
    const float ab = fabsf(b);
    return ab > 1e-15f && ab < 1e15f;   // false for NaN
}

template <bool FAST>
__device__ __forceinline__ float xdiv(float a, float b, float y) {
    if (!FAST) return a / b;
    float q = a * y;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// RN_f32(RN_f64(1 / s)) for a float s, as the reference's `1. / (w0/z0 + w1/z1 + w2/z2)` evaluates it (kernel.cu:68): the
// double rounding is harmless here -- a float midpoint m has 25 significant bits, so s*m is a 49-bit product and cannot
// lie within 2^-53 of 1 without being 1 -- hence the value is simply the correctly rounded fp32 reciprocal.
__device__ __forceinline__ float rcp_as_double(float s) { return 1.0f / s; }

// blockIdx -> (image, tile): consecutive logical ids live on the same XCD (observed placement:
// block b runs on XCD b % 8), so one image's faces are fetched into one L2.
__device__ __forceinline__ int logical_tile(const RasterArgs& a) {
    const int b = blockIdx.x;
    return (b & 7) * a.tiles_per_xcd + (b >> 3);
}

struct Pixel {
    int bn, pn;
    float xp, yp;
    bool valid;
};

__device__ __forceinline__ float ndc_coord(int i, int S) {
    // kernel.cu:345-346: evaluated in double, then narrowed
    return (float)((2. * i + 1. - S) / S);
}

__device__ __forceinline__ Pixel pixel_of_thread(const RasterArgs& a, int tile_id, int& tx0, int& ty0) {
    Pixel p;
    p.bn = tile_id / a.tiles_per_image;
    const int t = tile_id - p.bn * a.tiles_per_image;
    ty0 = (t / a.tiles_per_row) * TILE;
    tx0 = (t % a.tiles_per_row) * TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = tx0 + (wave & 1) * 8 + (lane & 7);
    const int row = ty0 + (wave >> 1) * 8 + (lane >> 3);
    p.valid = col < a.S && row < a.S;
    p.pn = row * a.S + col;
    p.xp = ndc_coord(col, a.S);
    p.yp = ndc_coord(a.S - 1 - row, a.S);
    return p;
}

// ------------------------------------------------------------------------------------------------
// per-face precompute (kernel.cu:245-305)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void face_setup_kernel(const float* __restrict__ faces,
                                                         float* __restrict__ info, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = faces + (size_t)i * 9;
    float* o = info + (size_t)i * 27;
    const float x0 = v[0], y0 = v[1], x1 = v[3], y1 = v[4], x2 = v[6], y2 = v[7];
    const float adj[9] = {y1 - y2, x2 - x1, x1 * y2 - x2 * y1,
                          y2 - y0, x0 - x2, x2 * y0 - x0 * y2,
                          y0 - y1, x1 - x0, x0 * y1 - x1 * y0};
    float det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
    det = det > 0 ? (float)fmax((double)det, 1e-10) : (float)fmin((double)det, -1e-10);
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = adj[k] / det;
    const float vv[9] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            o[9 + 3 * j + k] = vv[3 * j] * vv[3 * k] + vv[3 * j + 1] * vv[3 * k + 1] + 1;
    const float px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    bool found = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int p = (k + 1) % 3, q = (k + 2) % 3;
        const bool obt = (px[p] - px[k]) * (px[q] - px[k]) + (py[p] - py[k]) * (py[q] - py[k]) < 0;
        if (obt && !found) { o[18 + k] = 1; found = true; }
    }
}

// ------------------------------------------------------------------------------------------------
// tile binning: ordered compaction of the faces whose dilated bbox touches the tile rectangle
// ------------------------------------------------------------------------------------------------
struct TileRect { float x_lo, x_hi, y_lo, y_hi; };

__device__ __forceinline__ TileRect tile_rect(const RasterArgs& a, int tx0, int ty0) {
    TileRect r;
    const int xl = tx0, xh = min(tx0 + TILE - 1, a.S - 1);
    const int rl = ty0, rh = min(ty0 + TILE - 1, a.S - 1);
    r.x_lo = ndc_coord(xl, a.S);
    r.x_hi = ndc_coord(xh, a.S);
    r.y_hi = ndc_coord(a.S - 1 - rl, a.S);
    r.y_lo = ndc_coord(a.S - 1 - rh, a.S);
    return r;
}

// exact tile-level form of check_border (kernel.cu:32-38): the face is dropped for the tile only
// if every pixel centre of the tile fails the per-pixel test (the pixel coordinates are monotone)
__device__ __forceinline__ bool face_touches_tile(const float* v, float margin, const TileRect& r) {
    const float hx = max3f(v[0], v[3], v[6]) + margin, lx = min3f(v[0], v[3], v[6]) - margin;
    const float hy = max3f(v[1], v[4], v[7]) + margin, ly = min3f(v[1], v[4], v[7]) - margin;
    return !(r.x_lo > hx || r.x_hi < lx || r.y_lo > hy || r.y_hi < ly);
}

// appends (in order) the faces [c, c+256) that touch the tile; returns the new list length
__device__ __forceinline__ int bin_chunk(const RasterArgs& a, int bn, int c, const TileRect& rect,
                                         unsigned* list, int* wave_cnt, int n) {
    const int f = c + threadIdx.x;
    bool hit = false;
    if (f < a.F) hit = face_touches_tile(a.faces + ((size_t)bn * a.F + f) * 9, a.margin, rect);
    const unsigned long long m = __ballot(hit);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = n, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int cw = wave_cnt[w];
        if (w < wave) base += cw;
        total += cw;
    }
    if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)f;
    __syncthreads();
    return n + total;
}

// stages faces list[s .. s+cnt) into LDS records (8 threads per face) and hoists the face-only terms of the pair loop
template <bool WITH_TEX, bool WITH_TEX2 = false>
__device__ __forceinline__ void stage_faces(const RasterArgs& a, int bn, const unsigned* list, int s,
                                            int cnt, float* stage, float4* bbox) {
    const int slot = threadIdx.x >> 3, part = threadIdx.x & 7;
    if (slot < cnt) {
        const unsigned f = list[s + slot];
        const size_t g = (size_t)bn * a.F + f;
        const float* fv = a.faces + g * 9;
        const float* fi = a.faces_info + g * 27;
        float* rec = stage + slot * REC;
#pragma unroll
        for (int j0 = 0; j0 < 40; j0 += 8) {
            const int j = j0 + part;
            if (j < 9) rec[R_V + j] = fv[j];
            else if (j < 30) rec[R_INV + (j - 9)] = fi[j - 9];
            else if (j < 39) { if (WITH_TEX) rec[R_TEX + (j - 30)] = a.textures[g * 9 + (j - 30)]; }
        }
        if (WITH_TEX2) {
            rec[R_TEX2 + part] = a.textures2[g * 9 + part];
            if (part == 0) rec[R_TEX2 + 8] = a.textures2[g * 9 + 8];
        }
        if (part < 3) {
            // edge `part` runs from corner v0 = part to v1 = part + 1 (kernel.cu:41-45, func `closest point on an edge`)
            const int v0 = part, v1 = part == 2 ? 0 : part + 1;
            const float a0 = fi[9 + 3 * v0 + 0] - fi[9 + 3 * v1 + 0];
            const float a1 = fi[9 + 3 * v0 + 1] - fi[9 + 3 * v1 + 1];
            const float a2 = fi[9 + 3 * v0 + 2] - fi[9 + 3 * v1 + 2];
            rec[R_EA + 3 * part + 0] = a0;
            rec[R_EA + 3 * part + 1] = a1;
            rec[R_EA + 3 * part + 2] = a2;
            const float den = pick3(v0, a0, a1, a2) - pick3(v1, a0, a1, a2);
            rec[R_EDEN + part] = den;
            rec[R_ERCP + part] = 1.0f / den;
        } else if (part < 6) {
            const int k = part - 3;
            rec[R_ZRCP + k] = 1.0f / fv[3 * k + 2];
            // early-out threshold for barycentric weight k.  |grad w_k| = 1 / height_k, so w_k < -d * |grad w_k| puts the
            // pixel further than d behind the line through the opposite edge, i.e. at a true distance > d from the
            // triangle.  The reference skips a pair when ITS computed distance satisfies dis >= threshold (kernel.cu:384);
            // that distance is |sum_k (t_k - w_k) v_k| with t a convex combination on an edge, so it can fall short of the
            // true one only through the rounding of the w_k it is built from.  Bound E on that (u = 2^-24):
            //   w_k = inv_k . (x, y, 1):     3u * (|inv_k0| + |inv_k1| + |inv_k2|)          (products and sums)
            //   inv = adj / det:             |w_k| * 4u * D / |det|, D = sum |x_i| |y_j - y_k|   (det cancels for slivers)
            //                                2u * vmax^2 / |det|                                (adj_k2 = x_i y_j - x_j y_i)
            //   |w_k| <= 1 + |grad w_k| * (diameter + 2 * margin) for every pixel of the dilated bounding box
            // d = 1.02 * margin + 5 * E * max|v|.  For well-shaped faces that is 1-3 % above the margin; for slivers E
            // exceeds the margin and the early-out is effectively off; degenerate faces switch it off (+inf).
            const float x0 = fv[0], y0 = fv[1], x1 = fv[3], y1 = fv[4], x2 = fv[6], y2 = fv[7];
            const float e01 = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
            const float e12 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
            const float e20 = (x0 - x2) * (x0 - x2) + (y0 - y2) * (y0 - y2);
            const float det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0);
            const float dsum = fabsf(x2) * fabsf(y0 - y1) + fabsf(x0) * fabsf(y1 - y2) + fabsf(x1) * fabsf(y2 - y0);
            const float vmax = fmaxf(1.f, fmaxf(fmaxf(fmaxf(fabsf(x0), fabsf(y0)), fmaxf(fabsf(x1), fabsf(y1))), fmaxf(fabsf(x2), fabsf(y2))));
            const float reach = sqrtf(fmaxf(fmaxf(e01, e12), e20)) + 2.f * a.margin;
            const float rel_det = 2.4e-7f * dsum / fabsf(det);
            float err = 3.f * 1.2e-7f * vmax * vmax / fabsf(det);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const float gj = sqrtf(fi[3 * j] * fi[3 * j] + fi[3 * j + 1] * fi[3 * j + 1]);
                err += 1.8e-7f * (fabsf(fi[3 * j]) + fabsf(fi[3 * j + 1]) + fabsf(fi[3 * j + 2])) + (1.f + gj * reach) * rel_det;
            }
            const bool usable = a.dist_mode == SCP_DIST_EUCLIDEAN && fabsf(det) > 1e-9f && fminf(fminf(e01, e12), e20) > 0.f;
            const float gn = sqrtf(fi[3 * k] * fi[3 * k] + fi[3 * k + 1] * fi[3 * k + 1]);
            const float thr = (1.02f * a.margin + 5.f * err * vmax) * gn + 1e-6f;
            rec[R_PRE + k] = (usable && thr == thr) ? thr : INFINITY;
        } else if (part == 6) {
            bbox[slot] = make_float4(min3f(fv[0], fv[3], fv[6]) - a.margin, max3f(fv[0], fv[3], fv[6]) + a.margin,
                                     min3f(fv[1], fv[4], fv[7]) - a.margin, max3f(fv[1], fv[4], fv[7]) + a.margin);
        } else {
            bool ok = fast_div_range(fv[2]) && fast_div_range(fv[5]) && fast_div_range(fv[8]);
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const int v1 = e == 2 ? 0 : e + 1;
                const float av0 = fi[9 + 3 * e + e] - fi[9 + 3 * v1 + e];
                const float av1 = fi[9 + 3 * e + v1] - fi[9 + 3 * v1 + v1];
                ok = ok && fast_div_range(av0 - av1);
            }
            reinterpret_cast<unsigned*>(rec)[R_IDX] = f;
            reinterpret_cast<unsigned*>(rec)[R_SLOW] = ok ? 0u : 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per (pixel, face) geometry shared by forward and backward (kernel.cu:24-151, :375-404)
// `rec` may be a per-lane pointer (backward: one face per 16-lane row)
// ------------------------------------------------------------------------------------------------
struct Cover {
    float w[3], t[3];
    float sign, dx, dy, dis, frag;
};

__device__ __forceinline__ bool weights_inside(const float* w) {
    return w[0] <= 1 && w[0] >= 0 && w[1] <= 1 && w[1] >= 0 && w[2] <= 1 && w[2] >= 0;
}

__device__ __forceinline__ bool front_facing(const float* v) {
    return (v[7] - v[1]) * (v[3] - v[0]) < (v[4] - v[1]) * (v[6] - v[0]);
}

// kernel.cu:55-60.  The reference clamps in double (`max(min(w, 1.), 0.)`, `max(sum, 1e-5)`): on float operands a
// double min/max only selects one of them, and RN_f32(1e-5) is the largest float below the double 1e-5, so the float
// clamps below select exactly the same values.
__device__ __forceinline__ void clip_weights(float* w) {
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = fmaxf(fminf(w[k], 1.f), 0.f);
    const float s = fmaxf(w[0] + w[1] + w[2], 1e-5f);
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] /= s;
}

// parameter of the projection of the pixel onto edge e = (v0 = e, v1 = e + 1): (w . a - a[v1]) / (a[v0] - a[v1])
template <bool FAST>
__device__ __forceinline__ float edge_param(const float* rec, const float* w, int e) {
    const float* ea = rec + R_EA + 3 * e;
    const float a0 = ea[0], a1 = ea[1], a2 = ea[2];
    const float av1 = pick3(e == 2 ? 0 : e + 1, a0, a1, a2);
    return xdiv<FAST>(w[0] * a0 + w[1] * a1 + w[2] * a2 - av1, rec[R_EDEN + e], rec[R_ERCP + e]);
}

template <bool FAST>
__device__ __forceinline__ void euclid_distance(Cover& c, const float* v, const float* rec, float xp, float yp) {
    const float* w = c.w;
    if (w[0] > 0 && w[1] > 0 && w[2] > 0 && w[0] < 1 && w[1] < 1 && w[2] < 1) {
        float best = 100000000, bx = 0, by = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int v0 = k, v1 = (k + 1) % 3, v2 = (k + 2) % 3;
            float t0[3];
            t0[v0] = edge_param<FAST>(rec, w, k);
            t0[v1] = 1 - t0[v0];
            t0[v2] = 0;
            t0[0] -= w[0]; t0[1] -= w[1]; t0[2] -= w[2];
            const float ex = t0[0] * v[0] + t0[1] * v[3] + t0[2] * v[6];
            const float ey = t0[0] * v[1] + t0[1] * v[4] + t0[2] * v[7];
            const float d = ex * ex + ey * ey;
            if (d < best) { best = d; bx = ex; by = ey; c.t[0] = t0[0]; c.t[1] = t0[1]; c.t[2] = t0[2]; }
        }
        c.dx = bx; c.dy = by; c.sign = 1;
    } else {
        const float o0 = rec[R_OBT + 0], o1 = rec[R_OBT + 1], o2 = rec[R_OBT + 2];
        int v0 = 0;  // (the reference leaves v0 = -1 in a rounding corner case and reads out of
                     //  bounds; edge 0 is used there, like the oracle)
        if (w[1] <= 0 && w[2] <= 0) {
            v0 = 0;
            if (o0 == 1 && (xp - v[0]) * (v[6] - v[0]) + (yp - v[1]) * (v[7] - v[1]) > 0) v0 = 2;
        } else if (w[2] <= 0 && w[0] <= 0) {
            v0 = 1;
            if (o1 == 1 && (xp - v[3]) * (v[0] - v[3]) + (yp - v[4]) * (v[1] - v[4]) > 0) v0 = 0;
        } else if (w[0] <= 0 && w[1] <= 0) {
            v0 = 2;
            if (o2 == 1 && (xp - v[6]) * (v[3] - v[6]) + (yp - v[7]) * (v[4] - v[7]) > 0) v0 = 1;
        } else if (w[0] <= 0) v0 = 1;
        else if (w[1] <= 0) v0 = 2;
        else if (w[2] <= 0) v0 = 0;
        const int v1 = v0 == 2 ? 0 : v0 + 1;
        const float tp = edge_param<FAST>(rec, w, v0);
        const float tq = 1 - tp;
        float t[3];
        t[0] = v0 == 0 ? tp : (v1 == 0 ? tq : 0.f);
        t[1] = v0 == 1 ? tp : (v1 == 1 ? tq : 0.f);
        t[2] = v0 == 2 ? tp : (v1 == 2 ? tq : 0.f);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            t[k] = fminf(fmaxf(t[k], 0.f), 1.f);      // kernel.cu:141 clamps in double: a selection, see clip_weights
            t[k] -= w[k];
            c.t[k] = t[k];
        }
        c.dx = t[0] * v[0] + t[1] * v[3] + t[2] * v[6];
        c.dy = t[0] * v[1] + t[1] * v[4] + t[2] * v[7];
        c.sign = -1;
    }
}

// returns false when the pair is skipped (bbox is tested by the caller)
template <bool FAST>
__device__ __forceinline__ bool pair_coverage(const RasterArgs& a, Cover& c, const float* v,
                                              const float* rec, float xp, float yp) {
    const float* inv = rec + R_INV;
#pragma unroll
    for (int k = 0; k < 3; k++) c.w[k] = inv[3 * k] * xp + inv[3 * k + 1] * yp + inv[3 * k + 2];
    c.sign = 0; c.dx = 0; c.dy = 0; c.dis = 0;
    c.t[0] = c.t[1] = c.t[2] = 0;
    if (a.dist_mode == SCP_DIST_EUCLIDEAN) {
        euclid_distance<FAST>(c, v, rec, xp, yp);
        c.dis = c.dx * c.dx + c.dy * c.dy;
        if (c.sign < 0 && c.dis >= a.threshold) return false;
        c.frag = (float)(1. / (1. + (double)expf(xdiv<FAST>(-c.sign * c.dis, a.sigma, a.rcp_sigma))));
    } else if (a.dist_mode == SCP_DIST_BARYCENTRIC) {
        const float* w = c.w;
        float d = w[0] > w[1] ? (w[1] > w[2] ? w[2] : w[1]) : (w[0] > w[2] ? w[2] : w[0]);
        c.dis = d > 0 ? d * d : -(d * d);
        c.t[0] = w[0]; c.t[1] = w[1]; c.t[2] = w[2];
        if (-c.dis >= a.threshold) return false;
        c.frag = (float)(1. / (1. + (double)expf(xdiv<FAST>(-c.dis, a.sigma, a.rcp_sigma))));
    } else {
        c.frag = weights_inside(c.w) ? 1.f : 0.f;
        if (c.frag == 0.f) return false;
    }
    return true;
}

// perspective-correct depth of the pixel on the face (kernel.cu:66-69): 1 / (w0/z0 + w1/z1 + w2/z2), outer division in double
template <bool FAST>
__device__ __forceinline__ float pair_depth(const float* w, const float* v, const float* rec) {
    const float s = xdiv<FAST>(w[0], v[2], rec[R_ZRCP + 0]) + xdiv<FAST>(w[1], v[5], rec[R_ZRCP + 1]) +
                    xdiv<FAST>(w[2], v[8], rec[R_ZRCP + 2]);
    return rcp_as_double(s);
}

// surface-texture sampling reads global memory (kernel.cu:178-188); `limit` keeps the reference's
// out-of-range index (w == 1 exactly) inside the buffer
__device__ __forceinline__ int surface_texel(const float* w, int R) {
    const int wx = (int)(w[0] * R), wy = (int)(w[1] * R);
    if ((w[0] + w[1]) * R - wx - wy <= 1) return wy * R + wx;
    return (R - 1 - wy) * R + (R - 1 - wx);
}

template <int SAMPLE>
__device__ __forceinline__ float sample_colour(const RasterArgs& a, const float* rec, size_t face_g,
                                               const float* w, int k) {
    if (SAMPLE == SCP_SAMPLE_VERTEX) {
        const float* tex = rec + R_TEX;
        return w[0] * tex[k] + w[1] * tex[3 + k] + w[2] * tex[6 + k];
    } else {
        const size_t total = (size_t)a.B * a.F * a.T * 3;
        size_t idx = face_g * a.T * 3 + (size_t)(surface_texel(w, a.R) * 3 + k);
        if (idx >= total) idx = total - 1;
        return a.textures[idx];
    }
}

// ------------------------------------------------------------------------------------------------
// forward (kernel.cu:308-483)
// ------------------------------------------------------------------------------------------------
struct PixelState {            // the order-dependent per-pixel aggregates of kernel.cu:354-367
    float col[4];
    float sm_sum, sm_max, zmin;
    int fmin;
};

// bbox-surviving pixel outside the dilated triangle? (see R_PRE)  Same w as pair_coverage computes.
__device__ __forceinline__ bool behind_an_edge(const float* rec, float xp, float yp) {
    const float* inv = rec + R_INV;
    bool out = false;
#pragma unroll
    for (int k = 0; k < 3; k++) out = out || (inv[3 * k] * xp + inv[3 * k + 1] * yp + inv[3 * k + 2] < -rec[R_PRE + k]);
    return out;
}

struct HardState {             // z-buffer of the fused hard-colour output (dual forward)
    float col[3];
    float zmin;
    int fmin;
};

template <int RGB, int SAMPLE, bool FAST, bool DUAL>
__device__ __forceinline__ void forward_pair(const RasterArgs& a, PixelState& st, HardState& hs, const float* rec, float xp,
                                             float yp, int bn) {
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = rec[R_V + k];
    Cover cv;
    if (!pair_coverage<FAST>(a, cv, v, rec, xp, yp)) return;

    if (a.alpha_mode == SCP_ALPHA_PROD) st.col[3] = (float)((double)st.col[3] * (1. - (double)cv.frag));
    else if (a.alpha_mode == SCP_ALPHA_SUM) st.col[3] += cv.frag;
    else if (cv.frag > 0.5) st.col[3] = 1.f;

    float wc[3] = {cv.w[0], cv.w[1], cv.w[2]};
    clip_weights(wc);
    const float zp = pair_depth<FAST>(wc, v, rec);
    if (zp < a.near_ || zp > a.far_) return;

    const unsigned f = reinterpret_cast<const unsigned*>(rec)[R_IDX];
    const size_t face_g = (size_t)bn * a.F + f;
    if (DUAL) {      // the hard-colour pass of the same coverage (kernel.cu:428-439 with func_id_rgb = 0, vertex textures)
        if (zp < hs.zmin && weights_inside(cv.w) && (a.double_side || front_facing(v))) {
            hs.zmin = zp;
            hs.fmin = (int)f;
            const float* tex = rec + R_TEX2;
#pragma unroll
            for (int k = 0; k < 3; k++) hs.col[k] = wc[0] * tex[k] + wc[1] * tex[3 + k] + wc[2] * tex[6 + k];
        }
    }
    if (RGB == SCP_RGB_HARD) {
        if (zp < st.zmin && weights_inside(cv.w) && (a.double_side || front_facing(v))) {
            st.zmin = zp;
            st.fmin = (int)f;
#pragma unroll
            for (int k = 0; k < 3; k++) st.col[k] = sample_colour<SAMPLE>(a, rec, face_g, wc, k);
        }
    } else if (front_facing(v) || a.double_side) {
        const float zn = xdiv<FAST>(a.far_ - zp, a.range, a.rcp_range);
        float rescale = 1.f;
        if (zn > st.sm_max) {
            rescale = expf(xdiv<FAST>(st.sm_max - zn, a.gamma, a.rcp_gamma));
            st.sm_max = zn;
        }
        const float ez = expf(xdiv<FAST>(zn - st.sm_max, a.gamma, a.rcp_gamma));
        st.sm_sum = rescale * st.sm_sum + ez * cv.frag;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float ck = sample_colour<SAMPLE>(a, rec, face_g, wc, k);
            st.col[k] = rescale * st.col[k] + ez * cv.frag * ck;
        }
    }
}

template <int RGB, int SAMPLE, bool DUAL = false>
__global__ __launch_bounds__(THREADS) void raster_forward_kernel(const RasterArgs a) {
    __shared__ __attribute__((aligned(16))) float stage[NB * REC];
    __shared__ float4 bbox[NB];
    __shared__ unsigned list[LIST_CAP];
    __shared__ int wave_cnt[4];

    const int tile_id = logical_tile(a);
    if (tile_id >= a.total_tiles) return;
    int tx0, ty0;
    const Pixel px = pixel_of_thread(a, tile_id, tx0, ty0);
    const TileRect rect = tile_rect(a, tx0, ty0);
    const size_t npix = (size_t)a.S * a.S;
    float* out = a.soft_colors + (size_t)px.bn * 4 * npix + px.pn;

    PixelState st;
    st.col[0] = st.col[1] = st.col[2] = 1.f;
    st.col[3] = a.alpha_mode == SCP_ALPHA_PROD ? 1.f : 0.f;
    st.sm_sum = expf(a.eps / a.gamma);
    st.sm_max = a.eps;
    if (px.valid) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float bg = out[k * npix];
            st.col[k] = RGB == SCP_RGB_HARD ? bg : bg * st.sm_sum;
        }
    }
    st.zmin = 10000000;
    st.fmin = -1;
    HardState hs;
    hs.col[0] = hs.col[1] = hs.col[2] = 0.f;
    hs.zmin = 10000000;
    hs.fmin = -1;

    int n = 0;
    for (int c = 0; c < a.F; c += THREADS) {
        n = bin_chunk(a, px.bn, c, rect, list, wave_cnt, n);
        if (n <= LIST_CAP - THREADS && c + THREADS < a.F) continue;
        for (int s = 0; s < n; s += NB) {
            const int cnt = min(NB, n - s);
            stage_faces<SAMPLE == SCP_SAMPLE_VERTEX, DUAL>(a, px.bn, list, s, cnt, stage, bbox);
            __syncthreads();
            if (px.valid) {
                for (int q = 0; q < cnt; q++) {
                    const float4 bb = bbox[q];
                    if (px.xp > bb.y || px.xp < bb.x || px.yp > bb.w || px.yp < bb.z) continue;
                    const float* rec = stage + q * REC;
                    if (behind_an_edge(rec, px.xp, px.yp)) continue;
                    if (reinterpret_cast<const unsigned*>(rec)[R_SLOW] != 0u || a.const_slow)
                        forward_pair<RGB, SAMPLE, false, DUAL>(a, st, hs, rec, px.xp, px.yp, px.bn);
                    else
                        forward_pair<RGB, SAMPLE, true, DUAL>(a, st, hs, rec, px.xp, px.yp, px.bn);
                }
            }
            __syncthreads();
        }
        n = 0;
    }

    if (!px.valid) return;
    float alpha;
    if (a.alpha_mode == SCP_ALPHA_PROD) alpha = (float)(1. - (double)st.col[3]);
    else if (a.alpha_mode == SCP_ALPHA_SUM) alpha = st.col[3] / a.F;
    else alpha = st.col[3];
    out[3 * npix] = alpha;
    if (DUAL) {      // same coverage -> same alpha plane; colours only where a face won the z-test (kernel.cu:466-472)
        float* out2 = a.soft_colors2 + (size_t)px.bn * 4 * npix + px.pn;
        out2[3 * npix] = alpha;
        if (hs.fmin != -1) {
#pragma unroll
            for (int k = 0; k < 3; k++) out2[k * npix] = hs.col[k];
        }
        float* ag2 = a.aggrs_info2 + (size_t)px.bn * 2 * npix + px.pn;
        ag2[0] = hs.zmin;
        ag2[npix] = (float)hs.fmin;
    }

    float* ag = a.aggrs_info + (size_t)px.bn * 2 * npix + px.pn;
    if (RGB == SCP_RGB_HARD) {
        if (st.fmin != -1) {
#pragma unroll
            for (int k = 0; k < 3; k++) out[k * npix] = st.col[k];
        }
        ag[0] = st.zmin;
        ag[npix] = (float)st.fmin;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) out[k * npix] = st.col[k] / st.sm_sum;
        ag[0] = st.sm_sum;
        ag[npix] = st.sm_max;
    }
}

// ------------------------------------------------------------------------------------------------
// forward on a per-wavefront PAIR QUEUE (round 6 experiment, opt-in SCP_RASTER_FWD=pq; softmax rgb + vertex textures).  Built on the
// estimate of DESIGN 5.3 (r5), bit-identical to raster_forward_kernel -- and measured 1.2-1.4x SLOWER than it, see forward_legacy().
//
// raster_forward_kernel above runs the whole pair arithmetic under the divergence of its face loop: a face that covers 36 of a
// wavefront's 64 pixels costs the full ~270 instructions with 28 lanes idle (measured: lane use 56 %, profiles/r04_pmc_softras.txt).
// Only a small part of that arithmetic is order dependent.  Split it:
//   scan      (lane = pixel)  bbox + early-out test of every staged face, survivors appended IN (face, pixel) ORDER to a u16 ring,
//                             the face's hit mask and ring position kept per face;
//   coverage  (lane = PAIR)   as soon as 64 pairs are queued: barycentrics, distance, sigmoid, clipped weights, depth, normalised
//                             depth, interpolated colours -- ~200 of the ~270 instructions, at full lane use -- into a result ring;
//   apply     (lane = pixel)  the group's faces in order: a pixel reads ITS record of the face (position = the face's base + rank of
//                             the lane in the hit mask) and performs the order-dependent state update (alpha product in double, online
//                             softmax with rescale, z-test of the fused hard-colour output) exactly as forward_pair does.
// Same expressions, same order per pixel => the images are bit-identical to raster_forward_kernel's (tests/test_softras_gpu.py compares
// the two; SCP_RASTER_FWD=pq selects this kernel).  Everything the two phases exchange stays inside one wavefront: no barrier
// besides the two per staged batch that the old kernel has as well.
// ------------------------------------------------------------------------------------------------
constexpr int FQCAP = 128;                        // queue / result ring entries per wavefront (outstanding <= 63 + 64)
constexpr unsigned FP_COVER = 1u, FP_SOFT = 2u, FP_HARD = 4u;
// result record (dwords): 0 flags, 1 frag, 2 normalised depth, 3-5 colour; dual: 6 depth, 7-9 hard colour, 10 face index, 11 unused
template <bool DUAL> struct FwdRec { static constexpr int N = DUAL ? 12 : 6; };

template <bool FAST, bool DUAL>
__device__ __forceinline__ void forward_cover(const RasterArgs& a, const float* rec, float xp, float yp, float* out) {
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = rec[R_V + k];
    Cover cv;
    unsigned flags = 0u;
    float frag = 0.f, zn = 0.f, zp = 0.f, col[3] = {0.f, 0.f, 0.f}, hard[3] = {0.f, 0.f, 0.f};
    if (pair_coverage<FAST>(a, cv, v, rec, xp, yp)) {
        flags = FP_COVER;
        frag = cv.frag;
        float wc[3] = {cv.w[0], cv.w[1], cv.w[2]};
        clip_weights(wc);
        zp = pair_depth<FAST>(wc, v, rec);
        if (!(zp < a.near_ || zp > a.far_)) {
            const bool front = front_facing(v) || a.double_side;
            if (DUAL && front && weights_inside(cv.w)) {
                flags |= FP_HARD;
                const float* tex = rec + R_TEX2;
#pragma unroll
                for (int k = 0; k < 3; k++) hard[k] = wc[0] * tex[k] + wc[1] * tex[3 + k] + wc[2] * tex[6 + k];
            }
            if (front) {
                flags |= FP_SOFT;
                zn = xdiv<FAST>(a.far_ - zp, a.range, a.rcp_range);
                const float* tex = rec + R_TEX;
#pragma unroll
                for (int k = 0; k < 3; k++) col[k] = wc[0] * tex[k] + wc[1] * tex[3 + k] + wc[2] * tex[6 + k];
            }
        }
    }
    out[0] = __uint_as_float(flags);
    out[1] = frag;
    out[2] = zn;
    out[3] = col[0]; out[4] = col[1]; out[5] = col[2];
    if (DUAL) {
        out[6] = zp;
        out[7] = hard[0]; out[8] = hard[1]; out[9] = hard[2];
        out[10] = __uint_as_float(reinterpret_cast<const unsigned*>(rec)[R_IDX]);
    }
}

// the order-dependent half of forward_pair (kernel.cu:405-453), from a result record
template <bool FAST, bool DUAL>
__device__ __forceinline__ void forward_apply(const RasterArgs& a, PixelState& st, HardState& hs, const float* r) {
    const unsigned flags = __float_as_uint(r[0]);
    if (!(flags & FP_COVER)) return;
    const float frag = r[1];
    if (a.alpha_mode == SCP_ALPHA_PROD) st.col[3] = (float)((double)st.col[3] * (1. - (double)frag));
    else if (a.alpha_mode == SCP_ALPHA_SUM) st.col[3] += frag;
    else if (frag > 0.5) st.col[3] = 1.f;
    if (DUAL && (flags & FP_HARD)) {
        const float zp = r[6];
        if (zp < hs.zmin) {
            hs.zmin = zp;
            hs.fmin = (int)__float_as_uint(r[10]);
            hs.col[0] = r[7]; hs.col[1] = r[8]; hs.col[2] = r[9];
        }
    }
    if (flags & FP_SOFT) {
        const float zn = r[2];
        float rescale = 1.f;
        if (zn > st.sm_max) {
            rescale = expf(xdiv<FAST>(st.sm_max - zn, a.gamma, a.rcp_gamma));
            st.sm_max = zn;
        }
        const float ez = expf(xdiv<FAST>(zn - st.sm_max, a.gamma, a.rcp_gamma));
        st.sm_sum = rescale * st.sm_sum + ez * frag;
#pragma unroll
        for (int k = 0; k < 3; k++) st.col[k] = rescale * st.col[k] + ez * frag * r[3 + k];
    }
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool DUAL>
__global__ __launch_bounds__(THREADS) void raster_forward_pq_kernel(const RasterArgs a) {
    constexpr int NREC = FwdRec<DUAL>::N;
    __shared__ __attribute__((aligned(16))) float stage[NB * REC];
    __shared__ __attribute__((aligned(16))) float ring[4 * FQCAP * NREC];
    __shared__ float4 bbox[NB];
    __shared__ unsigned list[LIST_CAP];
    __shared__ unsigned long long hitmask[4 * NB];
    __shared__ int hitbase[4 * NB];
    __shared__ float2 pixxy[THREADS];
    __shared__ unsigned short queue[4 * FQCAP];
    __shared__ int wave_cnt[4];

    const int tile_id = logical_tile(a);
    if (tile_id >= a.total_tiles) return;
    int tx0, ty0;
    const Pixel px = pixel_of_thread(a, tile_id, tx0, ty0);
    const TileRect rect = tile_rect(a, tx0, ty0);
    const size_t npix = (size_t)a.S * a.S;
    float* out = a.soft_colors + (size_t)px.bn * 4 * npix + px.pn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;

    PixelState st;
    st.col[0] = st.col[1] = st.col[2] = 1.f;
    st.col[3] = a.alpha_mode == SCP_ALPHA_PROD ? 1.f : 0.f;
    st.sm_sum = expf(a.eps / a.gamma);
    st.sm_max = a.eps;
    if (px.valid) {
#pragma unroll
        for (int k = 0; k < 3; k++) st.col[k] = out[k * npix] * st.sm_sum;
    }
    st.zmin = 10000000;
    st.fmin = -1;
    HardState hs;
    hs.col[0] = hs.col[1] = hs.col[2] = 0.f;
    hs.zmin = 10000000;
    hs.fmin = -1;

    pixxy[threadIdx.x] = make_float2(px.xp, px.yp);
    unsigned short* qw = queue + wave * FQCAP;
    float* rw = ring + wave * FQCAP * NREC;
    unsigned long long* hm = hitmask + wave * NB;
    int* hb = hitbase + wave * NB;
    const float2* pxy = pixxy + wave * 64;

    int n = 0;
    for (int c = 0; c < a.F; c += THREADS) {
        n = bin_chunk(a, px.bn, c, rect, list, wave_cnt, n);
        if (n <= LIST_CAP - THREADS && c + THREADS < a.F) continue;
        for (int s = 0; s < n; s += NB) {
            const int cnt = min(NB, n - s);
            stage_faces<true, DUAL>(a, px.bn, list, s, cnt, stage, bbox);
            __syncthreads();
            int head = 0, tail = 0, q = 0;
            while (true) {
                // scan: pairs in (face, pixel) order; the face's mask and first ring position stay available for the apply phase
                while (tail - head < 64 && q < cnt) {
                    const float4 bb = bbox[q];
                    bool hit = px.valid && !(px.xp > bb.y || px.xp < bb.x || px.yp > bb.w || px.yp < bb.z);
                    if (__ballot(hit) != 0ull) hit = hit && !behind_an_edge(stage + q * REC, px.xp, px.yp);
                    const unsigned long long m = __ballot(hit);
                    if (lane == 0) { hm[q] = m; hb[q] = tail; }
                    if (m != 0ull) {
                        if (hit) qw[(tail + __popcll(m & lt)) & (FQCAP - 1)] = (unsigned short)(lane | (q << 8));
                        tail += __popcll(m);
                    }
                    q++;
                }
                if (tail == head) break;
                const int ng = min(64, tail - head);
                wave_sync();
                // coverage: lane = pair
                if (lane < ng) {
                    const int pos = (head + lane) & (FQCAP - 1);
                    const unsigned e = qw[pos];
                    const float* rec = stage + (e >> 8) * REC;
                    const float2 xy = pxy[e & 63u];
                    float* r = rw + pos * NREC;
                    if (reinterpret_cast<const unsigned*>(rec)[R_SLOW] != 0u || a.const_slow) forward_cover<false, DUAL>(a, rec, xy.x, xy.y, r);
                    else forward_cover<true, DUAL>(a, rec, xy.x, xy.y, r);
                }
                wave_sync();
                // apply: lane = pixel, the group's faces in order
                const int qf = __builtin_amdgcn_readfirstlane((int)(qw[head & (FQCAP - 1)] >> 8));
                const int ql = __builtin_amdgcn_readfirstlane((int)(qw[(head + ng - 1) & (FQCAP - 1)] >> 8));
                for (int qq = qf; qq <= ql; qq++) {
                    const unsigned long long m = hm[qq];
                    if (m == 0ull) continue;
                    const int pos = hb[qq] + __popcll(m & lt);
                    if (((m >> lane) & 1ull) && pos >= head && pos < head + ng) {
                        const float* r = rw + (pos & (FQCAP - 1)) * NREC;
                        if (a.const_slow) forward_apply<false, DUAL>(a, st, hs, r);
                        else forward_apply<true, DUAL>(a, st, hs, r);
                    }
                }
                head += ng;
                wave_sync();
            }
            __syncthreads();
        }
        n = 0;
    }

    if (!px.valid) return;
    float alpha;
    if (a.alpha_mode == SCP_ALPHA_PROD) alpha = (float)(1. - (double)st.col[3]);
    else if (a.alpha_mode == SCP_ALPHA_SUM) alpha = st.col[3] / a.F;
    else alpha = st.col[3];
    out[3 * npix] = alpha;
    if (DUAL) {
        float* out2 = a.soft_colors2 + (size_t)px.bn * 4 * npix + px.pn;
        out2[3 * npix] = alpha;
        if (hs.fmin != -1) {
#pragma unroll
            for (int k = 0; k < 3; k++) out2[k * npix] = hs.col[k];
        }
        float* ag2 = a.aggrs_info2 + (size_t)px.bn * 2 * npix + px.pn;
        ag2[0] = hs.zmin;
        ag2[npix] = (float)hs.fmin;
    }
    float* ag = a.aggrs_info + (size_t)px.bn * 2 * npix + px.pn;
#pragma unroll
    for (int k = 0; k < 3; k++) out[k * npix] = st.col[k] / st.sm_sum;
    ag[0] = st.sm_sum;
    ag[npix] = st.sm_max;
}

// ------------------------------------------------------------------------------------------------
// backward (kernel.cu:486-668)
// ------------------------------------------------------------------------------------------------
// Cross-lane reductions with DPP (data-parallel primitives: the cross-lane operand is read through
// the VALU's DPP path, no LDS traffic -- ds_bpermute based __shfl_xor made the reduction 60 % of the
// kernel).  DPP controls (gfx9 family): quad_perm [1,0,3,2] = lane^1, [2,3,0,1] = lane^2,
// row_half_mirror = lane^7 (within 8), row_mirror = lane^15 (within 16).
template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

template <int HALF, int CTRL>
__device__ __forceinline__ void scatter_step(float* g, int lane) {
    const bool up = (lane & HALF) != 0;   // the partner (lane ^ 15 / ^7 / ^2 / ^1) has this bit flipped
#pragma unroll
    for (int i = 0; i < HALF; i++) {
        const float keep = up ? g[i + HALF] : g[i];
        const float send = up ? g[i] : g[i + HALF];
        g[i] = keep + dpp_get<CTRL>(send);
    }
}

// reduce-scatter of NV (8 or 16) values over groups of NV lanes: afterwards lane l holds the sum over
// its NV-lane group of g[l & (NV-1)]
template <int NV>
__device__ __forceinline__ float group_reduce_scatter(float* g) {
    const int lane = threadIdx.x & 63;
    if (NV == 16) scatter_step<8, DPP_MIRROR>(g, lane);
    scatter_step<4, DPP_HALF_MIRROR>(g, lane);
    scatter_step<2, DPP_XOR2>(g, lane);
    scatter_step<1, DPP_XOR1>(g, lane);
    return g[0];
}

// sum over the 16-lane row, result in every lane of the row
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_get<DPP_XOR1>(x);
    x += dpp_get<DPP_XOR2>(x);
    x += dpp_get<DPP_HALF_MIRROR>(x);
    x += dpp_get<DPP_MIRROR>(x);
    return x;
}

// the 18 partial derivatives of one (pixel, face) pair (kernel.cu:560-660); g[] arrives zeroed
template <int RGB, int SAMPLE, bool FAST>
__device__ __forceinline__ void backward_pair(const RasterArgs& a, float* g, const float* rec, const float* pr, int bn) {
    const float xp = pr[P_XP], yp = pr[P_YP];
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = rec[R_V + k];
    Cover cv;
    if (!pair_coverage<FAST>(a, cv, v, rec, xp, yp)) return;
    const float img3 = pr[P_IMG + 3];
    float c_xy = 0;
    float c_alpha = pr[P_GIMG + 3];
    if (a.alpha_mode == SCP_ALPHA_SUM) c_alpha /= a.F;
    else if (a.alpha_mode == SCP_ALPHA_PROD)
        c_alpha = (float)((double)c_alpha * ((double)(1 - img3) / fmax((double)(1 - cv.frag), 1e-6)));
    c_xy += c_alpha;

    float w[3] = {cv.w[0], cv.w[1], cv.w[2]};
    clip_weights(w);
    const float zp = pair_depth<FAST>(w, v, rec);
    if (zp < a.near_ || zp > a.far_) return;
    const unsigned f = reinterpret_cast<const unsigned*>(rec)[R_IDX];
    const size_t face_g = (size_t)bn * a.F + f;
    const float sm_max = pr[P_MAX];
    if (RGB == SCP_RGB_HARD) {
        if ((float)(int)f == sm_max) {
            if (SAMPLE == SCP_SAMPLE_VERTEX) {
#pragma unroll
                for (int k = 0; k < 3; k++)
#pragma unroll
                    for (int j = 0; j < 3; j++) g[9 + 3 * j + k] = w[j] * pr[P_GIMG + k];
            } else {
                const int texel = surface_texel(w, a.R);
                if (texel >= 0 && texel < a.T) {
                    float* gt = a.grad_textures + face_g * a.T * 3 + texel * 3;
#pragma unroll
                    for (int k = 0; k < 3; k++) atomicAdd(gt + k, pr[P_GIMG + k]);
                }
            }
        }
    } else if (front_facing(v) || a.double_side) {
        float c_rgb = 0.f;
        const float zn = xdiv<FAST>(a.far_ - zp, a.range, a.rcp_range);
        const float ez = expf(xdiv<FAST>(zn - sm_max, a.gamma, a.rcp_gamma));
        const float zs = xdiv<FAST>(cv.frag * ez, pr[P_SUM], pr[P_RSUM]);
        int texel = 0;
        if (SAMPLE == SCP_SAMPLE_SURFACE) texel = surface_texel(w, a.R);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float gk = pr[P_GIMG + k];
            if (SAMPLE == SCP_SAMPLE_VERTEX) {
#pragma unroll
                for (int j = 0; j < 3; j++) g[9 + 3 * j + k] = zs * (w[j] * gk);
            } else if (texel >= 0 && texel < a.T) {
                atomicAdd(a.grad_textures + face_g * a.T * 3 + texel * 3 + k, zs * gk);
            }
            const float ck = sample_colour<SAMPLE>(a, rec, face_g, w, k);
            c_rgb += gk * (ck - pr[P_IMG + k]);
        }
        c_rgb *= zs;
        c_xy += c_rgb / cv.frag;
        const float c_z = xdiv<FAST>(xdiv<FAST>(c_rgb, a.gamma, a.rcp_gamma), a.nrange, a.rcp_nrange) * zp * zp;
        g[2] = xdiv<FAST>(xdiv<FAST>(c_z * w[0], v[2], rec[R_ZRCP + 0]), v[2], rec[R_ZRCP + 0]);
        g[5] = xdiv<FAST>(xdiv<FAST>(c_z * w[1], v[5], rec[R_ZRCP + 1]), v[5], rec[R_ZRCP + 1]);
        g[8] = xdiv<FAST>(xdiv<FAST>(c_z * w[2], v[8], rec[R_ZRCP + 2]), v[8], rec[R_ZRCP + 2]);
    }

    c_xy *= xdiv<FAST>(cv.frag * (1 - cv.frag), a.sigma, a.rcp_sigma);
    if (a.dist_mode == SCP_DIST_EUCLIDEAN) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            g[3 * k + 0] = 2 * cv.sign * c_xy * (cv.t[k] + cv.w[k]) * cv.dx;
            g[3 * k + 1] = 2 * cv.sign * c_xy * (cv.t[k] + cv.w[k]) * cv.dy;
        }
    } else if (a.dist_mode == SCP_DIST_BARYCENTRIC) {
        // kernel.cu:161-175
        const float* t = cv.t;
        const float* inv = rec + R_INV;
        const int pm = t[0] > t[1] ? (t[1] > t[2] ? 2 : 1) : (t[0] > t[2] ? 2 : 0);
        const float scale2 = cv.dis > 0 ? sqrtf(cv.dis) : sqrtf(-cv.dis);
#pragma unroll
        for (int l = 0; l < 2; l++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float gk = 0;
#pragma unroll
                for (int qq = 0; qq < 3; qq++)
                    gk += -inv[3 * pm + l] * inv[3 * k + qq] * (qq == 0 ? xp : (qq == 1 ? yp : 1.f));
                float r = gk * c_xy;
                r = (float)((double)r * (2. * (double)scale2));
                g[3 * k + l] = r;
            }
    }
}

// One group of 64 queued (pixel, face) pairs: lane = pair.  Queue entry: bits 0-7 pixel of the wavefront's quadrant,
// bits 8-12 staged-face slot, bit 15 = padding lane (carries the slot of its row).  Every 16-lane row holds one face.
// NG = 18: [grad_face 0..8 | grad_texture 0..8] (softmax rgb or vertex textures);
// NG = 6 : grad_face x/y only (hard rgb with surface textures, e.g. the mask pass)
template <int RGB, int SAMPLE>
__device__ __forceinline__ void backward_group(const RasterArgs& a, const float* stage, const float* pix_wave,
                                               float* acc, const unsigned short* queue, int head, int n_valid,
                                               int bn) {
    constexpr bool FULL = !(RGB == SCP_RGB_HARD && SAMPLE == SCP_SAMPLE_SURFACE);
    constexpr int NACC = 18;
    const int lane = threadIdx.x & 63;
    const unsigned e = queue[(head + lane) & (QCAP - 1)];
    const bool in_range = lane < n_valid;
    const bool live = in_range && !(e & 0x8000u);
    const unsigned long long row_first = __ballot(in_range);          // rows are whole: bit of the row's first lane
    const int q = in_range ? (int)((e >> 8) & 31u) : 0;
    const float* rec = stage + q * REC;
    const float* pr = pix_wave + (live ? (int)(e & 0xFFu) : 0) * PIXREC;

    float g[18];
#pragma unroll
    for (int k = 0; k < 18; k++) g[k] = 0.f;
    if (live) {
        const bool slow = reinterpret_cast<const unsigned*>(rec)[R_SLOW] != 0u ||
                          reinterpret_cast<const unsigned*>(pr)[P_SLOW] != 0u || a.const_slow;
        if (slow) backward_pair<RGB, SAMPLE, false>(a, g, rec, pr, bn);
        else backward_pair<RGB, SAMPLE, true>(a, g, rec, pr, bn);
    }
    // rows are merged by the LDS atomics themselves (several rows of a group may hold the same face)
    const bool row_live = (row_first >> (lane & 48)) & 1ull;
    float* slot_acc = acc + q * NACC;
    if (FULL) {
        const float g16 = row_sum16(g[16]), g17 = row_sum16(g[17]);
        const float r = group_reduce_scatter<16>(g);
        const int l16 = lane & 15;
        if (row_live) {
            atomicAdd(slot_acc + l16, r);
            if (l16 < 2) atomicAdd(slot_acc + 16 + l16, l16 == 0 ? g16 : g17);
        }
    } else {
        float h[8] = {g[0], g[1], g[3], g[4], g[6], g[7], 0.f, 0.f};
        const float r = group_reduce_scatter<8>(h);
        const int l8 = lane & 7;
        if (row_live && l8 < 6) atomicAdd(slot_acc + (l8 + (l8 >> 1)), r);  // 0,1,3,4,6,7
    }
}

template <int RGB, int SAMPLE>
__global__ __launch_bounds__(THREADS) void raster_backward_kernel(const RasterArgs a) {
    constexpr int NACC = 18;
    __shared__ __attribute__((aligned(16))) float stage[NB * REC];
    __shared__ __attribute__((aligned(16))) float pix[THREADS * PIXREC];
    __shared__ float4 bbox[NB];
    __shared__ float acc[NB * NACC];
    __shared__ unsigned list[LIST_CAP];
    __shared__ unsigned short queue[4 * QCAP];
    __shared__ int wave_cnt[4];

    const int tile_id = logical_tile(a);
    if (tile_id >= a.total_tiles) return;
    int tx0, ty0;
    const Pixel px = pixel_of_thread(a, tile_id, tx0, ty0);
    const TileRect rect = tile_rect(a, tx0, ty0);
    const size_t npix = (size_t)a.S * a.S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    {   // this thread's pixel -> LDS pixel table (entry threadIdx.x = wave * 64 + lane)
        float* pr = pix + threadIdx.x * PIXREC;
        float sm_sum = 1.f;
        if (px.valid) {
            const float* ip = a.soft_colors_in + (size_t)px.bn * 4 * npix + px.pn;
            const float* gp = a.grad_soft_colors + (size_t)px.bn * 4 * npix + px.pn;
#pragma unroll
            for (int k = 0; k < 4; k++) { pr[P_IMG + k] = ip[k * npix]; pr[P_GIMG + k] = gp[k * npix]; }
            sm_sum = a.aggrs_info_in[((size_t)px.bn * 2 + 0) * npix + px.pn];
            pr[P_MAX] = a.aggrs_info_in[((size_t)px.bn * 2 + 1) * npix + px.pn];
        }
        pr[P_SUM] = sm_sum;
        pr[P_RSUM] = 1.0f / sm_sum;
        pr[P_XP] = px.xp;
        pr[P_YP] = px.yp;
        reinterpret_cast<unsigned*>(pr)[P_SLOW] = fast_div_range(sm_sum) ? 0u : 1u;
    }
    for (int i = threadIdx.x; i < NB * NACC; i += THREADS) acc[i] = 0.f;
    unsigned short* qw = queue + wave * QCAP;
    const float* pix_wave = pix + wave * 64 * PIXREC;

    int n = 0;
    for (int c = 0; c < a.F; c += THREADS) {
        n = bin_chunk(a, px.bn, c, rect, list, wave_cnt, n);
        if (n <= LIST_CAP - THREADS && c + THREADS < a.F) continue;
        for (int s = 0; s < n; s += NB) {
            const int cnt = min(NB, n - s);
            stage_faces<SAMPLE == SCP_SAMPLE_VERTEX>(a, px.bn, list, s, cnt, stage, bbox);
            __syncthreads();
            // scan: the surviving pairs of this wavefront's 64 pixels go to the ring face by face, each face's run padded
            // to whole 16-lane rows; a group of 64 is processed as soon as it exists (and the remainder at the end)
            int head = 0, tail = 0, q = 0;
            while (true) {
                while (tail - head < 64 && q < cnt) {
                    const float4 bb = bbox[q];
                    bool hit = px.valid && !(px.xp > bb.y || px.xp < bb.x || px.yp > bb.w || px.yp < bb.z);
                    if (__ballot(hit) != 0ull) {
                        hit = hit && !behind_an_edge(stage + q * REC, px.xp, px.yp);
                        const unsigned long long m = __ballot(hit);
                        if (m != 0ull) {
                            const int c_hit = __popcll(m);
                            if (hit) qw[(tail + __popcll(m & ((1ull << lane) - 1ull))) & (QCAP - 1)] = (unsigned short)(lane | (q << 8));
                            const int pad = (-(tail + c_hit)) & 15;
                            if (lane < pad) qw[(tail + c_hit + lane) & (QCAP - 1)] = (unsigned short)(0x8000 | (q << 8));
                            tail += c_hit + pad;
                        }
                    }
                    q++;
                }
                if (tail == head) break;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const int n_valid = min(64, tail - head);
                backward_group<RGB, SAMPLE>(a, stage, pix_wave, acc, qw, head, n_valid, px.bn);
                head += n_valid;
            }
            __syncthreads();
            // flush the per-batch accumulators: one global atomic per (tile, face, component)
            for (int i = threadIdx.x; i < cnt * NACC; i += THREADS) {
                const int qq = i / NACC, j = i - qq * NACC;
                const float val = acc[i];
                if (val != 0.f) {
                    const unsigned f = reinterpret_cast<const unsigned*>(stage + qq * REC)[R_IDX];
                    const size_t face_g = (size_t)px.bn * a.F + f;
                    if (j < 9) atomicAdd(a.grad_faces + face_g * 9 + j, val);
                    else atomicAdd(a.grad_textures + face_g * 9 + (j - 9), val);
                    acc[i] = 0.f;
                }
            }
            __syncthreads();
        }
        n = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// instrumentation: bbox-surviving (pixel, face) pairs
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void count_pairs_kernel(const RasterArgs a) {
    __shared__ float4 bbox[NB];
    __shared__ unsigned list[LIST_CAP];
    __shared__ int wave_cnt[4];
    const int tile_id = logical_tile(a);
    if (tile_id >= a.total_tiles) return;
    int tx0, ty0;
    const Pixel px = pixel_of_thread(a, tile_id, tx0, ty0);
    const TileRect rect = tile_rect(a, tx0, ty0);
    unsigned long long mine = 0;
    int n = 0;
    for (int c = 0; c < a.F; c += THREADS) {
        n = bin_chunk(a, px.bn, c, rect, list, wave_cnt, n);
        if (n <= LIST_CAP - THREADS && c + THREADS < a.F) continue;
        for (int s = 0; s < n; s += NB) {
            const int cnt = min(NB, n - s);
            if ((int)threadIdx.x < cnt) {
                const float* fv = a.faces + ((size_t)px.bn * a.F + list[s + threadIdx.x]) * 9;
                bbox[threadIdx.x] = make_float4(min3f(fv[0], fv[3], fv[6]) - a.margin, max3f(fv[0], fv[3], fv[6]) + a.margin,
                                                min3f(fv[1], fv[4], fv[7]) - a.margin, max3f(fv[1], fv[4], fv[7]) + a.margin);
            }
            __syncthreads();
            if (px.valid)
                for (int q = 0; q < cnt; q++) {
                    const float4 bb = bbox[q];
                    mine += !(px.xp > bb.y || px.xp < bb.x || px.yp > bb.w || px.yp < bb.z);
                }
            __syncthreads();
        }
        n = 0;
    }
    // wavefront sum, one atomic per wavefront
    unsigned lo = (unsigned)mine;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) lo += __shfl_xor(lo, m);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.counter, (unsigned long long)lo);
}

// self-test of xdiv: n (a, b) pairs from a counter-based generator, mantissas uniform, exponents of b in +-2^40 and of a
// in +-2^60 around it, a share of adversarial mantissas (all ones / one / powers of two); counts xdiv(a,b,RN(1/b)) != a/b
__global__ void exact_division_selftest_kernel(unsigned long long n, unsigned seed, unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        unsigned ma = (unsigned)x & 0x7FFFFFu, mb = (unsigned)(x >> 23) & 0x7FFFFFu;
        const unsigned kind = (unsigned)(x >> 46) & 15u;
        if (kind == 0) mb = 0x7FFFFFu; else if (kind == 1) mb = 0u; else if (kind == 2) mb = 1u; else if (kind == 3) ma = 0x7FFFFFu;
        else if (kind == 4) { ma = 0u; } else if (kind == 5) { mb = 0x7FFFFEu; ma = 1u; }
        const int eb = 127 + (int)((x >> 50) % 81u) - 40;
        const int ea = eb + (int)((x >> 57) % 121u) - 60;
        const float b = __uint_as_float(((unsigned)(x >> 63) << 31) | ((unsigned)eb << 23) | mb);
        const float av = __uint_as_float((((unsigned)(x >> 62) & 1u) << 31) | ((unsigned)max(1, min(ea, 254)) << 23) | ma);
        const float y = 1.0f / b;
        const float q = fast_div_range(b) ? xdiv<true>(av, b, y) : xdiv<false>(av, b, y);
        const float r = av / b;
        bad += __float_as_uint(q) != __float_as_uint(r);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int fill_args(RasterArgs& a, const scp_raster_params* p) {
    if (!p) return scp::fail(hipErrorInvalidValue, "scp_raster_params is NULL");
    if (p->batch_size < 0 || p->num_faces < 0 || p->image_size <= 0 || p->texture_size <= 0)
        return scp::fail(hipErrorInvalidValue, "bad sizes in scp_raster_params");
    if (p->func_id_dist < 0 || p->func_id_dist > 2 || p->func_id_rgb < 0 || p->func_id_rgb > 1 ||
        p->func_id_alpha < 0 || p->func_id_alpha > 2 || p->texture_sample_type < 0 || p->texture_sample_type > 1)
        return scp::fail(hipErrorInvalidValue, "unknown func_id / texture_sample_type");
    if (p->texture_sample_type == SCP_SAMPLE_VERTEX && p->texture_size != 3)
        return scp::fail(hipErrorInvalidValue, "vertex textures need texture_size == 3");
    a.B = p->batch_size; a.F = p->num_faces; a.S = p->image_size; a.T = p->texture_size;
    a.R = (int)sqrt((double)p->texture_size);
    a.tiles_per_row = (a.S + TILE - 1) / TILE;
    a.tiles_per_image = a.tiles_per_row * a.tiles_per_row;
    a.total_tiles = a.B * a.tiles_per_image;
    a.tiles_per_xcd = (a.total_tiles + 7) / 8;
    a.near_ = p->near_; a.far_ = p->far_; a.eps = p->eps; a.sigma = p->sigma_val;
    a.dist_eps = p->dist_eps; a.gamma = p->gamma_val;
    a.threshold = p->dist_eps * p->sigma_val;  // kernel.cu:352 (fp32 product)
    a.margin = sqrtf(a.threshold);             // kernel.cu:375 (fp32 sqrt)
    a.range = a.far_ - a.near_;                // kernel.cu:423 forms (far - near) in fp32
    a.nrange = a.near_ - a.far_;               // kernel.cu:637
    a.rcp_sigma = 1.0f / a.sigma; a.rcp_gamma = 1.0f / a.gamma;
    a.rcp_range = 1.0f / a.range; a.rcp_nrange = 1.0f / a.nrange;
    a.const_slow = !(fast_div_range(a.sigma) && fast_div_range(a.gamma) && fast_div_range(a.range));
    a.dist_mode = p->func_id_dist; a.alpha_mode = p->func_id_alpha; a.double_side = p->double_side != 0;
    return 0;
}

template <template <int, int> class Launcher>
int dispatch(const RasterArgs& a, int rgb, int sample, hipStream_t st) {
    const dim3 grid(a.tiles_per_xcd * 8), block(THREADS);
    if (rgb == SCP_RGB_HARD && sample == SCP_SAMPLE_SURFACE) Launcher<0, 0>::go(grid, block, st, a);
    else if (rgb == SCP_RGB_HARD) Launcher<0, 1>::go(grid, block, st, a);
    else if (sample == SCP_SAMPLE_SURFACE) Launcher<1, 0>::go(grid, block, st, a);
    else Launcher<1, 1>::go(grid, block, st, a);
    return scp::check_launch("soft_rasterize");
}

// SCP_RASTER_FWD=pq: the pair-queue forward kernel for the softmax / vertex-texture passes (A/B and the bit-equality test).  NOT the
// default: measured on the MI355X it is bit-identical to the per-face kernel and SLOWER (sigma = 1e-3 pass 0.95 vs 0.67 ms, sigma = 1e-4
// pass 0.41 vs 0.35 ms at B = 32, gpurun_out/r06i -> profiles/r06_softras_forward_ab.txt): the LDS round trip of the records and the
// face-ordered apply loop cost more than the full-lane coverage arithmetic saves.
bool forward_legacy() {
    const char* e = getenv("SCP_RASTER_FWD");
    return !(e != nullptr && strcmp(e, "pq") == 0);
}

template <int RGB, int SAMPLE> struct FwdLaunch {
    static void go(dim3 g, dim3 b, hipStream_t st, const RasterArgs& a) {
        hipLaunchKernelGGL((raster_forward_kernel<RGB, SAMPLE>), g, b, 0, st, a);
    }
};
template <int RGB, int SAMPLE> struct BwdLaunch {
    static void go(dim3 g, dim3 b, hipStream_t st, const RasterArgs& a) {
        hipLaunchKernelGGL((raster_backward_kernel<RGB, SAMPLE>), g, b, 0, st, a);
    }
};

}  // namespace

extern "C" int scp_soft_rasterize_forward(const float* faces, const float* textures, float* faces_info,
                                          float* aggrs_info, float* soft_colors,
                                          const scp_raster_params* p, void* stream) {
    RasterArgs a{};
    if (int e = fill_args(a, p)) return e;
    if (a.B == 0 || a.total_tiles == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    a.faces = faces; a.textures = textures; a.faces_info = faces_info;
    a.aggrs_info = aggrs_info; a.soft_colors = soft_colors;
    const int nf = a.B * a.F;
    if (nf > 0) {
        hipLaunchKernelGGL(face_setup_kernel, dim3((nf + 255) / 256), dim3(256), 0, st, faces, faces_info, nf);
        if (int e = scp::check_launch("face_setup")) return e;
    }
    if (p->func_id_rgb == SCP_RGB_SOFTMAX && p->texture_sample_type == SCP_SAMPLE_VERTEX && !forward_legacy()) {
        hipLaunchKernelGGL((raster_forward_pq_kernel<false>), dim3(a.tiles_per_xcd * 8), dim3(THREADS), 0, st, a);
        return scp::check_launch("soft_rasterize_forward (pair queue)");
    }
    return dispatch<FwdLaunch>(a, p->func_id_rgb, p->texture_sample_type, st);
}

extern "C" int scp_soft_rasterize_forward_dual(const float* faces, const float* textures, float* faces_info,
                                               float* aggrs_info, float* soft_colors, const float* textures_hard,
                                               float* aggrs_info_hard, float* soft_colors_hard,
                                               const scp_raster_params* p, void* stream) {
    RasterArgs a{};
    if (int e = fill_args(a, p)) return e;
    if (p->func_id_rgb != SCP_RGB_SOFTMAX || p->texture_sample_type != SCP_SAMPLE_VERTEX)
        return scp::fail(hipErrorInvalidValue, "forward_dual: the primary pass must be softmax rgb with vertex textures");
    if (a.B == 0 || a.total_tiles == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    a.faces = faces; a.textures = textures; a.faces_info = faces_info;
    a.aggrs_info = aggrs_info; a.soft_colors = soft_colors;
    a.textures2 = textures_hard; a.aggrs_info2 = aggrs_info_hard; a.soft_colors2 = soft_colors_hard;
    const int nf = a.B * a.F;
    if (nf > 0) {
        hipLaunchKernelGGL(face_setup_kernel, dim3((nf + 255) / 256), dim3(256), 0, st, faces, faces_info, nf);
        if (int e = scp::check_launch("face_setup")) return e;
    }
    if (forward_legacy())
        hipLaunchKernelGGL((raster_forward_kernel<SCP_RGB_SOFTMAX, SCP_SAMPLE_VERTEX, true>), dim3(a.tiles_per_xcd * 8),
                           dim3(THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((raster_forward_pq_kernel<true>), dim3(a.tiles_per_xcd * 8), dim3(THREADS), 0, st, a);
    return scp::check_launch("soft_rasterize_forward_dual");
}

extern "C" int scp_soft_rasterize_backward(const float* faces, const float* textures,
                                           const float* soft_colors, const float* faces_info,
                                           const float* aggrs_info, float* grad_faces,
                                           float* grad_textures, const float* grad_soft_colors,
                                           const scp_raster_params* p, void* stream) {
    RasterArgs a{};
    if (int e = fill_args(a, p)) return e;
    if (a.B == 0 || a.F == 0 || a.total_tiles == 0) return 0;
    a.faces = faces; a.textures = textures; a.faces_info = faces_info;
    a.soft_colors_in = soft_colors; a.aggrs_info_in = aggrs_info;
    a.grad_faces = grad_faces; a.grad_textures = grad_textures; a.grad_soft_colors = grad_soft_colors;
    return dispatch<BwdLaunch>(a, p->func_id_rgb, p->texture_sample_type, static_cast<hipStream_t>(stream));
}

extern "C" int scp_soft_rasterize_count_pairs(const float* faces, unsigned long long* count,
                                              const scp_raster_params* p, void* stream) {
    RasterArgs a{};
    if (int e = fill_args(a, p)) return e;
    if (a.B == 0 || a.F == 0) return 0;
    a.faces = faces; a.counter = count;
    hipLaunchKernelGGL(count_pairs_kernel, dim3(a.tiles_per_xcd * 8), dim3(THREADS), 0,
                       static_cast<hipStream_t>(stream), a);
    return scp::check_launch("count_pairs");
}

extern "C" int scp_selftest_exact_division(unsigned long long n, unsigned seed, unsigned long long* mismatches,
                                           void* stream) {
    hipLaunchKernelGGL(exact_division_selftest_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), n, seed,
                       mismatches);
    return scp::check_launch("exact_division_selftest");
}
