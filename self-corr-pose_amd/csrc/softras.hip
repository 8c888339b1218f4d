// self-corr-pose_amd/csrc/softras.hip -- SoftRas soft rasteriser for MI355X (gfx950), forward + backward.
//
// What it computes is fixed by the reference kernels
//   third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu
//     :245-305 per-face precompute, :308-483 per-pixel forward, :486-668 per-pixel backward
// (semantics checklist: SURVEY.md Appendix A).  How it computes it is CDNA4-first:
//
//   * one 256-thread workgroup (4 wavefronts of 64) owns a 16x16 pixel tile; each wavefront owns
//     an 8x8 quadrant so that a face whose dilated bounding box misses the quadrant costs the
//     wavefront a single LDS broadcast read + compare (s_cbranch_execz skips the rest);
//   * the tile first bins the faces: 256 faces per round are tested against the tile rectangle
//     and compacted IN FACE-INDEX ORDER (ballot + popcount prefix) into an LDS list -- order
//     matters, the alpha product / online softmax / z-buffer tie-break are order dependent and
//     must equal the reference's brute-force loop bit for bit;
//   * listed faces are staged 32 at a time into LDS as 44-float records (dilated bbox, corners,
//     inverse, Gram matrix, obtuse flags, vertex colours); all 64 lanes read the same record ->
//     LDS broadcast, no bank conflicts, nothing is re-read from L2 per pixel (the reference
//     re-reads every face from global memory in every thread);
//   * backward: per (wavefront, face) the 18 partial derivatives are combined with a butterfly
//     reduce-scatter over the wavefront (17+12 cross-lane ops instead of 18x6), added to an LDS
//     accumulator with ds_add_f32, and flushed with ONE global atomic per (tile, face, component)
//     -- the reference issues 9-18 global atomics per (pixel, face);
//   * blockIdx is remapped so that the tiles of one image run on one XCD (its faces stay in that
//     XCD's L2).
//
// Numerics: this file is compiled with -ffp-contract=off and without fast-math; every expression
// keeps the reference's evaluation order and its fp64 promotions, so the only
// source of difference from the CPU oracle is the last-ulp behaviour of expf.
#include <hip/hip_runtime.h>

#include "scp_hip.h"
#include "scp_common.h"

namespace {

// Division inside pure GRADIENT arithmetic (never in a coverage / clipping decision): SCP_FAST_GRAD_DIV selects v_rcp_f32 * x
// (1-2 ulp) instead of the IEEE sequence (~10 instructions); off by default, the oracle comparison then holds to the last bit
#ifdef SCP_FAST_GRAD_DIV
#define GDIV(x, y) ((x) * __builtin_amdgcn_rcpf(y))
#else
#define GDIV(x, y) ((x) / (y))
#endif

constexpr int TILE = 16;        // pixels per tile edge
constexpr int THREADS = 256;    // 4 wavefronts
constexpr int NB = 32;          // faces staged per round
constexpr int REC = 44;         // floats per staged face record
constexpr int LIST_CAP = 1024;  // binned face ids held before a flush

// record layout (floats)
constexpr int R_BBOX = 0;   // lo_x, hi_x, lo_y, hi_y  (already dilated by sqrt(threshold))
constexpr int R_V = 4;      // 9 corner coordinates
constexpr int R_INV = 13;   // 9
constexpr int R_SYM = 22;   // 9
constexpr int R_OBT = 31;   // 3
constexpr int R_TEX = 34;   // 9 (vertex colours) -- only valid when texture_size == 3
constexpr int R_IDX = 43;   // face index (int bits)

struct RasterArgs {
    const float* faces;
    const float* textures;
    const float* faces_info;
    float* aggrs_info;
    float* soft_colors;
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
    int dist_mode, alpha_mode, double_side;
    int dbg;   // profiling ablations (tools/softras_ablate.py): 1 = skip reduction+flush, 2 = skip pair math
};

int g_debug_flags = 0;

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float pick3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }

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

// stages faces list[s .. s+cnt) into LDS records; 8 threads per face
template <bool WITH_TEX>
__device__ __forceinline__ void stage_faces(const RasterArgs& a, int bn, const unsigned* list, int s,
                                            int cnt, float* stage) {
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
        if (part == 7) {
            rec[R_BBOX + 0] = min3f(fv[0], fv[3], fv[6]) - a.margin;
            rec[R_BBOX + 1] = max3f(fv[0], fv[3], fv[6]) + a.margin;
            rec[R_BBOX + 2] = min3f(fv[1], fv[4], fv[7]) - a.margin;
            rec[R_BBOX + 3] = max3f(fv[1], fv[4], fv[7]) + a.margin;
            reinterpret_cast<unsigned*>(rec)[R_IDX] = f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per (pixel, face) geometry shared by forward and backward (kernel.cu:24-151, :375-404)
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

__device__ __forceinline__ void clip_weights(float* w) {
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] = (float)fmax(fmin((double)w[k], 1.), 0.);
    const float s = (float)fmax((double)(w[0] + w[1] + w[2]), 1e-5);
#pragma unroll
    for (int k = 0; k < 3; k++) w[k] /= s;
}

// parameter of the projection of the pixel onto edge (v0 -> v1); `sym` points into the LDS record
// and may be indexed with a per-lane v0 (distinct rows sit in distinct banks)
__device__ __forceinline__ float edge_param(const float* sym, const float* w, int v0, int v1) {
    const float a0 = sym[3 * v0 + 0] - sym[3 * v1 + 0];
    const float a1 = sym[3 * v0 + 1] - sym[3 * v1 + 1];
    const float a2 = sym[3 * v0 + 2] - sym[3 * v1 + 2];
    const float av1 = pick3(v1, a0, a1, a2), av0 = pick3(v0, a0, a1, a2);
    return (w[0] * a0 + w[1] * a1 + w[2] * a2 - av1) / (av0 - av1);
}

__device__ __forceinline__ void euclid_distance(Cover& c, const float* v, const float* rec, float xp,
                                                float yp) {
    const float* sym = rec + R_SYM;
    const float* w = c.w;
    if (w[0] > 0 && w[1] > 0 && w[2] > 0 && w[0] < 1 && w[1] < 1 && w[2] < 1) {
        float best = 100000000, bx = 0, by = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int v0 = k, v1 = (k + 1) % 3, v2 = (k + 2) % 3;
            float t0[3];
            t0[v0] = edge_param(sym, w, v0, v1);
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
        const float tp = edge_param(sym, w, v0, v1);
        const float tq = 1 - tp;
        float t[3];
        t[0] = v0 == 0 ? tp : (v1 == 0 ? tq : 0.f);
        t[1] = v0 == 1 ? tp : (v1 == 1 ? tq : 0.f);
        t[2] = v0 == 2 ? tp : (v1 == 2 ? tq : 0.f);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            t[k] = (float)fmin(fmax((double)t[k], 0.), 1.);
            t[k] -= w[k];
            c.t[k] = t[k];
        }
        c.dx = t[0] * v[0] + t[1] * v[3] + t[2] * v[6];
        c.dy = t[0] * v[1] + t[1] * v[4] + t[2] * v[7];
        c.sign = -1;
    }
}

// returns false when the pair is skipped (bbox is tested by the caller)
__device__ __forceinline__ bool pair_coverage(const RasterArgs& a, Cover& c, const float* v,
                                              const float* rec, float xp, float yp) {
    const float* inv = rec + R_INV;
#pragma unroll
    for (int k = 0; k < 3; k++) c.w[k] = inv[3 * k] * xp + inv[3 * k + 1] * yp + inv[3 * k + 2];
    c.sign = 0; c.dx = 0; c.dy = 0; c.dis = 0;
    c.t[0] = c.t[1] = c.t[2] = 0;
    if (a.dist_mode == SCP_DIST_EUCLIDEAN) {
        euclid_distance(c, v, rec, xp, yp);
        c.dis = c.dx * c.dx + c.dy * c.dy;
        if (c.sign < 0 && c.dis >= a.threshold) return false;
        c.frag = (float)(1. / (1. + (double)expf(-c.sign * c.dis / a.sigma)));
    } else if (a.dist_mode == SCP_DIST_BARYCENTRIC) {
        const float* w = c.w;
        float d = w[0] > w[1] ? (w[1] > w[2] ? w[2] : w[1]) : (w[0] > w[2] ? w[2] : w[0]);
        c.dis = d > 0 ? d * d : -(d * d);
        c.t[0] = w[0]; c.t[1] = w[1]; c.t[2] = w[2];
        if (-c.dis >= a.threshold) return false;
        c.frag = (float)(1. / (1. + (double)expf(-c.dis / a.sigma)));
    } else {
        c.frag = weights_inside(c.w) ? 1.f : 0.f;
        if (c.frag == 0.f) return false;
    }
    return true;
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
template <int RGB, int SAMPLE>
__global__ __launch_bounds__(THREADS) void raster_forward_kernel(const RasterArgs a) {
    __shared__ __attribute__((aligned(16))) float stage[NB * REC];
    __shared__ unsigned list[LIST_CAP];
    __shared__ int wave_cnt[4];

    const int tile_id = logical_tile(a);
    if (tile_id >= a.total_tiles) return;
    int tx0, ty0;
    const Pixel px = pixel_of_thread(a, tile_id, tx0, ty0);
    const TileRect rect = tile_rect(a, tx0, ty0);
    const size_t npix = (size_t)a.S * a.S;
    float* out = a.soft_colors + (size_t)px.bn * 4 * npix + px.pn;

    float col[4] = {1.f, 1.f, 1.f, 0.f};
    if (a.alpha_mode == SCP_ALPHA_PROD) col[3] = 1.f;
    float sm_sum = expf(a.eps / a.gamma);
    float sm_max = a.eps;
    if (px.valid) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float bg = out[k * npix];
            col[k] = RGB == SCP_RGB_HARD ? bg : bg * sm_sum;
        }
    }
    float zmin = 10000000;
    int fmin = -1;

    int n = 0;
    for (int c = 0; c < a.F; c += THREADS) {
        n = bin_chunk(a, px.bn, c, rect, list, wave_cnt, n);
        if (n <= LIST_CAP - THREADS && c + THREADS < a.F) continue;
        for (int s = 0; s < n; s += NB) {
            const int cnt = min(NB, n - s);
            stage_faces<SAMPLE == SCP_SAMPLE_VERTEX>(a, px.bn, list, s, cnt, stage);
            __syncthreads();
            if (px.valid) {
                for (int q = 0; q < cnt; q++) {
                    const float* rec = stage + q * REC;
                    const float4 bb = *reinterpret_cast<const float4*>(rec + R_BBOX);
                    if (px.xp > bb.y || px.xp < bb.x || px.yp > bb.w || px.yp < bb.z) continue;
                    float v[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = rec[R_V + k];
                    Cover cv;
                    if (!pair_coverage(a, cv, v, rec, px.xp, px.yp)) continue;

                    if (a.alpha_mode == SCP_ALPHA_PROD) col[3] = (float)((double)col[3] * (1. - (double)cv.frag));
                    else if (a.alpha_mode == SCP_ALPHA_SUM) col[3] += cv.frag;
                    else if (cv.frag > 0.5) col[3] = 1.f;

                    float wc[3] = {cv.w[0], cv.w[1], cv.w[2]};
                    clip_weights(wc);
                    const float zp = (float)(1. / (double)(wc[0] / v[2] + wc[1] / v[5] + wc[2] / v[8]));
                    if (zp < a.near_ || zp > a.far_) continue;

                    const unsigned f = reinterpret_cast<const unsigned*>(rec)[R_IDX];
                    const size_t face_g = (size_t)px.bn * a.F + f;
                    if (RGB == SCP_RGB_HARD) {
                        if (zp < zmin && weights_inside(cv.w) && (a.double_side || front_facing(v))) {
                            zmin = zp;
                            fmin = (int)f;
#pragma unroll
                            for (int k = 0; k < 3; k++) col[k] = sample_colour<SAMPLE>(a, rec, face_g, wc, k);
                        }
                    } else {
                        if (front_facing(v) || a.double_side) {
                            const float zn = (a.far_ - zp) / (a.far_ - a.near_);
                            float rescale = 1.f;
                            if (zn > sm_max) {
                                rescale = expf((sm_max - zn) / a.gamma);
                                sm_max = zn;
                            }
                            const float ez = expf((zn - sm_max) / a.gamma);
                            sm_sum = rescale * sm_sum + ez * cv.frag;
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                const float ck = sample_colour<SAMPLE>(a, rec, face_g, wc, k);
                                col[k] = rescale * col[k] + ez * cv.frag * ck;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        n = 0;
    }

    if (!px.valid) return;
    if (a.alpha_mode == SCP_ALPHA_PROD) out[3 * npix] = (float)(1. - (double)col[3]);
    else if (a.alpha_mode == SCP_ALPHA_SUM) out[3 * npix] = col[3] / a.F;
    else out[3 * npix] = col[3];

    float* ag = a.aggrs_info + (size_t)px.bn * 2 * npix + px.pn;
    if (RGB == SCP_RGB_HARD) {
        if (fmin != -1) {
#pragma unroll
            for (int k = 0; k < 3; k++) out[k * npix] = col[k];
        }
        ag[0] = zmin;
        ag[npix] = (float)fmin;
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) out[k * npix] = col[k] / sm_sum;
        ag[0] = sm_sum;
        ag[npix] = sm_max;
    }
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

// NG = 18: [grad_face 0..8 | grad_texture 0..8] (softmax rgb or vertex textures);
// NG = 6 : grad_face x/y only (hard rgb with surface textures, e.g. the mask pass)
template <int RGB, int SAMPLE>
__global__ __launch_bounds__(THREADS) void raster_backward_kernel(const RasterArgs a) {
    constexpr bool FULL = !(RGB == SCP_RGB_HARD && SAMPLE == SCP_SAMPLE_SURFACE);
    constexpr int NACC = 18;
    __shared__ __attribute__((aligned(16))) float stage[NB * REC];
    __shared__ float acc[NB * NACC];
    __shared__ unsigned list[LIST_CAP];
    __shared__ int wave_cnt[4];

    const int tile_id = logical_tile(a);
    if (tile_id >= a.total_tiles) return;
    int tx0, ty0;
    const Pixel px = pixel_of_thread(a, tile_id, tx0, ty0);
    const TileRect rect = tile_rect(a, tx0, ty0);
    const size_t npix = (size_t)a.S * a.S;
    const int lane = threadIdx.x & 63;

    float img[4] = {0, 0, 0, 0}, gimg[4] = {0, 0, 0, 0}, sm_sum = 1.f, sm_max = 0.f;
    if (px.valid) {
        const float* ip = a.soft_colors_in + (size_t)px.bn * 4 * npix + px.pn;
        const float* gp = a.grad_soft_colors + (size_t)px.bn * 4 * npix + px.pn;
#pragma unroll
        for (int k = 0; k < 4; k++) { img[k] = ip[k * npix]; gimg[k] = gp[k * npix]; }
        sm_sum = a.aggrs_info_in[((size_t)px.bn * 2 + 0) * npix + px.pn];
        sm_max = a.aggrs_info_in[((size_t)px.bn * 2 + 1) * npix + px.pn];
    }
    for (int i = threadIdx.x; i < NB * NACC; i += THREADS) acc[i] = 0.f;

    int n = 0;
    for (int c = 0; c < a.F; c += THREADS) {
        n = bin_chunk(a, px.bn, c, rect, list, wave_cnt, n);
        if (n <= LIST_CAP - THREADS && c + THREADS < a.F) continue;
        for (int s = 0; s < n; s += NB) {
            const int cnt = min(NB, n - s);
            stage_faces<SAMPLE == SCP_SAMPLE_VERTEX>(a, px.bn, list, s, cnt, stage);
            __syncthreads();
            for (int q = 0; q < cnt; q++) {
                const float* rec = stage + q * REC;
                const float4 bb = *reinterpret_cast<const float4*>(rec + R_BBOX);
                float g[18];
#pragma unroll
                for (int k = 0; k < 18; k++) g[k] = 0.f;
                bool active = false;
                if (px.valid && !(a.dbg & 2) && !(px.xp > bb.y || px.xp < bb.x || px.yp > bb.w || px.yp < bb.z)) {
                    float v[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = rec[R_V + k];
                    Cover cv;
                    if (pair_coverage(a, cv, v, rec, px.xp, px.yp)) {
                        float c_xy = 0;
                        float c_alpha = gimg[3];
                        if (a.alpha_mode == SCP_ALPHA_SUM) c_alpha /= a.F;
                        else if (a.alpha_mode == SCP_ALPHA_PROD)
                            c_alpha = (float)((double)c_alpha *
                                              ((double)(1 - img[3]) / fmax((double)(1 - cv.frag), 1e-6)));
                        c_xy += c_alpha;

                        float w[3] = {cv.w[0], cv.w[1], cv.w[2]};
                        clip_weights(w);
                        const float zp = (float)(1. / (double)(w[0] / v[2] + w[1] / v[5] + w[2] / v[8]));
                        if (!(zp < a.near_ || zp > a.far_)) {
                            active = true;
                            const unsigned f = reinterpret_cast<const unsigned*>(rec)[R_IDX];
                            const size_t face_g = (size_t)px.bn * a.F + f;
                            if (RGB == SCP_RGB_HARD) {
                                if ((float)(int)f == sm_max) {
                                    if (SAMPLE == SCP_SAMPLE_VERTEX) {
#pragma unroll
                                        for (int k = 0; k < 3; k++)
#pragma unroll
                                            for (int j = 0; j < 3; j++) g[9 + 3 * j + k] = w[j] * gimg[k];
                                    } else {
                                        const int texel = surface_texel(w, a.R);
                                        if (texel >= 0 && texel < a.T) {
                                            float* gt = a.grad_textures + face_g * a.T * 3 + texel * 3;
#pragma unroll
                                            for (int k = 0; k < 3; k++) atomicAdd(gt + k, gimg[k]);
                                        }
                                    }
                                }
                            } else if (front_facing(v) || a.double_side) {
                                float c_rgb = 0.f;
                                const float zn = GDIV(a.far_ - zp, a.far_ - a.near_);
                                const float zs = GDIV(cv.frag * expf(GDIV(zn - sm_max, a.gamma)), sm_sum);
                                int texel = 0;
                                if (SAMPLE == SCP_SAMPLE_SURFACE) texel = surface_texel(w, a.R);
#pragma unroll
                                for (int k = 0; k < 3; k++) {
                                    const float gk = gimg[k];
                                    if (SAMPLE == SCP_SAMPLE_VERTEX) {
#pragma unroll
                                        for (int j = 0; j < 3; j++) g[9 + 3 * j + k] = zs * (w[j] * gk);
                                    } else if (texel >= 0 && texel < a.T) {
                                        atomicAdd(a.grad_textures + face_g * a.T * 3 + texel * 3 + k, zs * gk);
                                    }
                                    const float ck = sample_colour<SAMPLE>(a, rec, face_g, w, k);
                                    c_rgb += gk * (ck - img[k]);
                                }
                                c_rgb *= zs;
                                c_xy += GDIV(c_rgb, cv.frag);
                                const float c_z = GDIV(GDIV(c_rgb, a.gamma), a.near_ - a.far_) * zp * zp;
                                g[2] = GDIV(GDIV(c_z * w[0], v[2]), v[2]);
                                g[5] = GDIV(GDIV(c_z * w[1], v[5]), v[5]);
                                g[8] = GDIV(GDIV(c_z * w[2], v[8]), v[8]);
                            }

                            c_xy *= GDIV(cv.frag * (1 - cv.frag), a.sigma);
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
                                            gk += -inv[3 * pm + l] * inv[3 * k + qq] * (qq == 0 ? px.xp : (qq == 1 ? px.yp : 1.f));
                                        float r = gk * c_xy;
                                        r = (float)((double)r * (2. * (double)scale2));
                                        g[3 * k + l] = r;
                                    }
                            }
                        }
                    }
                }
                // wavefront-uniform: nothing to add if no lane produced a term
                if (__ballot(active) == 0ull) continue;
                if (a.dbg & 1) { if (g[0] + g[5] + g[11] == 12345.f) acc[0] = 1.f; continue; }
                float* slot_acc = acc + q * NACC;
                // the 4 rows (8 groups) of the wavefront are merged by the LDS atomics themselves
                if (FULL) {
                    const float g16 = row_sum16(g[16]), g17 = row_sum16(g[17]);
                    const float r = group_reduce_scatter<16>(g);
                    const int l16 = lane & 15;
                    atomicAdd(slot_acc + l16, r);
                    if (l16 < 2) atomicAdd(slot_acc + 16 + l16, l16 == 0 ? g16 : g17);
                } else {
                    float h[8] = {g[0], g[1], g[3], g[4], g[6], g[7], 0.f, 0.f};
                    const float r = group_reduce_scatter<8>(h);
                    const int l8 = lane & 7;
                    if (l8 < 6) atomicAdd(slot_acc + (l8 + (l8 >> 1)), r);  // 0,1,3,4,6,7
                }
            }
            __syncthreads();
            // flush the per-batch accumulators: one global atomic per (tile, face, component)
            for (int i = threadIdx.x; i < cnt * NACC; i += THREADS) {
                const int q = i / NACC, j = i - q * NACC;
                const float val = acc[i];
                if (val != 0.f) {
                    const unsigned f = reinterpret_cast<const unsigned*>(stage + q * REC)[R_IDX];
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
    __shared__ __attribute__((aligned(16))) float stage[NB * REC];
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
            // only the bbox part of the record is needed, but faces_info may be absent here
            const int slot = threadIdx.x >> 3, part = threadIdx.x & 7;
            if (slot < cnt && part == 0) {
                const float* fv = a.faces + ((size_t)px.bn * a.F + list[s + slot]) * 9;
                float* rec = stage + slot * REC;
                rec[0] = min3f(fv[0], fv[3], fv[6]) - a.margin;
                rec[1] = max3f(fv[0], fv[3], fv[6]) + a.margin;
                rec[2] = min3f(fv[1], fv[4], fv[7]) - a.margin;
                rec[3] = max3f(fv[1], fv[4], fv[7]) + a.margin;
            }
            __syncthreads();
            if (px.valid)
                for (int q = 0; q < cnt; q++) {
                    const float4 bb = *reinterpret_cast<const float4*>(stage + q * REC);
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
    a.dist_mode = p->func_id_dist; a.alpha_mode = p->func_id_alpha; a.double_side = p->double_side != 0;
    a.dbg = g_debug_flags;
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

extern "C" void scpdbg_set_flags(int flags) { g_debug_flags = flags; }

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
    return dispatch<FwdLaunch>(a, p->func_id_rgb, p->texture_sample_type, st);
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
