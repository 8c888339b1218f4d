// self-corr-pose_amd/csrc/corr_fused.hip -- feature <-> vertex correspondence (a7) without the score tensor.
//
// Replaces model/module/correspondence.py:42-53 (all-pairs scores, mask, softmax over pixels and over vertices, both
// soft-argmaxes) together with the 2x2 pooling of the scores that pretrained_corr.py:120-123 applies to them, forward and
// backward.  The reference materialises pc[B,4096,V] (0.34-1.3 GB), runs ~10 passes over it per direction and a
// [B,4096,V,3] temporary; round 1 here still wrote the scores (rocBLAS), re-read them twice, re-wrote them masked and went
// through them again in backward (~4 GB per step).  Here S = mesh_feat . img_feat is a K = 64 MFMA tile that never leaves
// the registers:
//
//   forward  (grid: 2-image-row strips x images): per 32-vertex tile, S^T[v][p] on the matrix cores (lane = pixel,
//            registers = vertices) ->  row softmax over vertices is lane local (online, like flash attention) and gives
//            match = softmax_V(tau_img S) @ verts;  column statistics over the strip's 128 pixels by DPP reductions
//            (max, sum e, sum e*gx, sum e*gy per vertex) -> per-strip partials, merged by a small kernel into
//            imatch = grid @ softmax_P(tau_mesh S);  the 2x2 pixel pooling is a quad DPP sum -> pooled[B,1024,V].
//            Written: pooled scores (the only form pretrained_corr.py consumes), match, imatch, row / column statistics.
//   backward (recompute): d S = tau_i P_r (g_match . verts_v - g_match . match_p) + tau_m P_c (g_imatch_v . grid_p -
//            g_imatch_v . imatch_v) + g_pooled / 4, zero on masked pixels, built per tile from the saved statistics and used
//            AS IS as the B operand of a second MFMA (the accumulator layout of v_mfma_f32_32x32x2 is its B layout):
//            kernel A (lane = pixel) contracts over vertices -> g_img_feat; kernel B (lane = vertex) over pixels ->
//            g_mesh_feat.  Deterministic: no atomics, fixed merge orders.
//
// Restrictions (else scp_amd/ops.py keeps the unfused kernels of corr.hip): C = 64 features, 64-pixel-wide feature map,
// even height.  Roofline: HBM for the record (algorithmic bytes = inputs + pooled + outputs, SURVEY 8d: ~0.13 GB at M1)
// but the kernels are VALU / MFMA bound (~30 VALU instructions per score for the two softmaxes).
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C = 64;            // feature channels
constexpr int WF = 64;           // feature-map width
// a forward block owns 2 image rows = 128 pixels = 32 pooled pixels
constexpr float MASKED_SCORE = -1e5f;
constexpr int MROW = 68;         // LDS row stride of a [32 vertices][64 channels] tile (conflict-free b128 reads)

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
// partner 16 lanes of the same 32-lane half (ds_swizzle bit mode: and 0x1F, or 0, xor 0x10)
__device__ __forceinline__ float swz16(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F)); }
__device__ __forceinline__ float other_half(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum32(float x) {       // sum over the 32 lanes of the lane's half, result in all of them
    x += dpp_get<DPP_XOR1>(x);
    x += dpp_get<DPP_XOR2>(x);
    x += dpp_get<DPP_HALF_MIRROR>(x);
    x += dpp_get<DPP_MIRROR>(x);
    return x + swz16(x);
}
__device__ __forceinline__ float max32(float x) {
    x = fmaxf(x, dpp_get<DPP_XOR1>(x));
    x = fmaxf(x, dpp_get<DPP_XOR2>(x));
    x = fmaxf(x, dpp_get<DPP_HALF_MIRROR>(x));
    x = fmaxf(x, dpp_get<DPP_MIRROR>(x));
    return fmaxf(x, swz16(x));
}

struct FvmArgs {
    const float* img;      // [B, 64, P]
    const float* mesh;     // [B, V, 64]
    const float* mask;     // [B, P]
    const float* verts;    // [B, V, 3]
    const float* grid;     // [2, P]
    float tau_img, tau_mesh;
    int B, P, V, hf, ntile, nblk;
    // forward outputs
    float* pooled;         // [B, P/4, V]
    float* match;          // [B, P, 3]
    float* rowstat;        // [B, P, 2]   (max of tau_img * s over vertices, sum of exp)
    float* colpart;        // [B, nblk, V, 4] (max, sum e, sum e gx, sum e gy) of tau_mesh * s over the strip's pixels
    const float* grid_half;// [2, P/4] or null: pixel grid at the pooled resolution -> also the column statistics of the POOLED scores
    float* colpart2;       // [B, nblk, V, 4] the same four numbers of tau_mesh * pooled over the strip's 32 pool cells (grid_half)
    // backward inputs
    const float* colstat;  // [B, V, 2]
    const float* imatch;   // [B, 2, V]
    const float* g_match;  // [B, P, 3]
    const float* g_imatch; // [B, 2, V]
    const float* g_pooled; // [B, P/4, V]
    const float* match_in; // [B, P, 3]
    const float* rowstat_in;
    float* g_img;          // [B, 64, P]
    float* g_mesh;         // [B, V, 64]
};

// blockIdx -> (unit, image), XCD-aware: workgroup i runs on XCD i % 8 (observed placement), so workgroup i takes logical id
// (i % 8) * per_xcd + i / 8 -- consecutive logical ids = the units of ONE image share an XCD, and that image's features / scores are
// fetched into one L2 instead of eight (round 6: the vertex-side backward re-read every image's [64, P] features on every XCD;
// 854 MB counter traffic for 165 MB algorithmic, profiles/r05_traffic.json).  Returns false for the padding workgroups.
__device__ __forceinline__ bool xcd_unit(int per_image, int B, int& unit, int& b) {
    const int total = per_image * B, per_xcd = (total + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || logical >= total) return false;
    b = logical / per_image;
    unit = logical - b * per_image;
    return true;
}
__host__ inline unsigned xcd_grid(int per_image, int B) { return 8u * (unsigned)((per_image * B + 7) >> 3); }

// pixel of lane l31 in wavefront w of strip blk: 16 x-positions x 2 rows, the four lanes 4g..4g+3 form one 2x2 pool cell
__device__ __forceinline__ int strip_pixel(int blk, int wave, int l31) {
    return (2 * blk + (l31 & 1)) * WF + 16 * wave + (l31 >> 1);
}

// stage vertices [v0, v0+32) of image b: features -> mt[32][MROW], coordinates -> vt[32][4]; rows past V are zero
__device__ __forceinline__ void stage_vertex_tile(const FvmArgs& a, int b, int v0, float* mt, float* vt) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i;               // float4 slot: row e >> 4, chunk e & 15
        const int row = e >> 4, ch = e & 15;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v0 + row < a.V) val = *reinterpret_cast<const float4*>(a.mesh + ((size_t)b * a.V + v0 + row) * C + 4 * ch);
        *reinterpret_cast<float4*>(mt + row * MROW + 4 * ch) = val;
    }
    if (tid < 32) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v0 + tid < a.V) {
            const float* vp = a.verts + ((size_t)b * a.V + v0 + tid) * 3;
            val = make_float4(vp[0], vp[1], vp[2], 0.f);
        }
        *reinterpret_cast<float4*>(vt + 4 * tid) = val;
    }
}

// S^T tile: acc[r] = sum_c mesh[v = acc_row(r, half)][c] * img[c][p(lane)];  breg[4t + r] = img[8t + 4 half + r][p]
__device__ __forceinline__ f32x16 score_tile(const float* mt, const float* breg, int l31, int half) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const float* mrow = mt + l31 * MROW + 4 * half;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float4 m = *reinterpret_cast<const float4*>(mrow + 8 * t);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(m.x, breg[4 * t + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(m.y, breg[4 * t + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(m.z, breg[4 * t + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(m.w, breg[4 * t + 3], acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ void load_pixel_operand(const FvmArgs& a, int b, int p, int half, float* breg) {
    const float* ip = a.img + (size_t)b * C * a.P + p;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) breg[4 * t + r] = ip[(size_t)(8 * t + 4 * half + r) * a.P];
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
// The second orientation of a 32 x 32 score tile: from accT[r] = S[v = acc_row(r, half)][p = l31] (lane = pixel, what the MFMA
// produces with the vertex tile as A operand) to accP[r] = S[p = acc_row(r, half)][v = l31] (lane = vertex) through a wavefront-
// private LDS tile (row stride 33: both the row-wise writes and the column-wise reads are conflict-free).  Each orientation makes
// one of the two softmaxes lane local; 32 LDS instructions instead of a second set of 32 MFMAs (2048 matrix-pipe cycles).
constexpr int TROW = 33;
__device__ __forceinline__ f32x16 transpose_tile(const f32x16& accT, float* tt, int l31, int half) {
#pragma unroll
    for (int r = 0; r < 16; r++) tt[acc_row(r, half) * TROW + l31] = accT[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 accP;
#pragma unroll
    for (int r = 0; r < 16; r++) accP[r] = tt[l31 * TROW + acc_row(r, half)];
    __builtin_amdgcn_wave_barrier();
    return accP;
}

__global__ __launch_bounds__(256) void fvm_forward_kernel(const FvmArgs a) {
    __shared__ __attribute__((aligned(16))) float mt[2][32 * MROW];
    __shared__ __attribute__((aligned(16))) float vt[2][32 * 4];
    __shared__ __attribute__((aligned(16))) float colred[2][4][32][4];
    __shared__ __attribute__((aligned(8))) float pixgrid[4][32][2];          // per wavefront pixel: grid x, y
    __shared__ float ttile[4][32 * TROW];                                     // per wavefront: the score tile on its way to the other orientation
    __shared__ __attribute__((aligned(16))) float colred2[2][4][32][4];       // column partials of the pooled scores (a9's bridge)

    int blk, b;
    if (!xcd_unit(a.nblk, a.B, blk, b)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int p = strip_pixel(blk, wave, l31);
    const bool masked = !(a.mask[(size_t)b * a.P + p] > 0.f);
    float breg[32];
    load_pixel_operand(a, b, p, half, breg);
    if (half == 0) *reinterpret_cast<float2*>(&pixgrid[wave][l31][0]) = make_float2(a.grid[p], a.grid[a.P + p]);
    // pixels of the lane = vertex orientation: register r <-> the wavefront's pixel acc_row(r, half); their mask bits
    const unsigned long long mball = __ballot(masked && half == 0);          // bit l31 = pixel l31 of this wavefront is masked
    unsigned mbits = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) mbits |= (unsigned)((mball >> acc_row(r, half)) & 1ull) << r;

    float m_run = -INFINITY, l_run = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;     // row softmax over vertices (this half's share)
    // pooled cells of the lane = vertex orientation: registers 4j..4j+3 are the 2x2 cell 2j + half of the wavefront
    float* pooled_col = a.pooled + ((size_t)b * (a.P / 4) + blk * (WF / 2) + 8 * wave + half) * a.V + l31;
    // pretrained_corr.py:123-126 takes a softmax over the POOLED pixels of every vertex' score map and its grid-weighted mean
    // (the "mesh -> image" half of the vertex bridge): the pooled values are in this lane's registers anyway, so their column
    // statistics are produced here as well (same partial / merge protocol as imatch) and the a9 pass over pooled[] disappears
    const bool bridge = a.grid_half != nullptr;
    float ghx[4] = {0.f, 0.f, 0.f, 0.f}, ghy[4] = {0.f, 0.f, 0.f, 0.f};
    if (bridge) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p2 = blk * (WF / 2) + 8 * wave + 2 * j + half;
            ghx[j] = a.grid_half[p2];
            ghy[j] = a.grid_half[a.P / 4 + p2];
        }
    }

    stage_vertex_tile(a, b, 0, mt[0], vt[0]);
    __syncthreads();
    for (int tv = 0; tv < a.ntile; tv++) {
        const int buf = tv & 1, v0 = 32 * tv;
        if (tv + 1 < a.ntile) stage_vertex_tile(a, b, v0 + 32, mt[buf ^ 1], vt[buf ^ 1]);
        f32x16 acc = score_tile(mt[buf], breg, l31, half);
        f32x16 accP = transpose_tile(acc, ttile[wave], l31, half);
        // ---- row softmax (lane = pixel, lane local): running max / sum / weighted vertex sum over this half's 16 vertices
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool valid = v0 + acc_row(r, half) < a.V;
            const float s = masked ? MASKED_SCORE : acc[r];
            acc[r] = s;
            if (valid) tmax = fmaxf(tmax, a.tau_img * s);
        }
        if (tmax > m_run) {
            const float sc = __expf(m_run - tmax);        // exp(-inf) = 0 the first time
            l_run *= sc; a0 *= sc; a1 *= sc; a2 *= sc;
            m_run = tmax;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int vr = acc_row(r, half);
            if (v0 + vr < a.V) {
                const float e = __expf(a.tau_img * acc[r] - m_run);
                const float4 vv = *reinterpret_cast<const float4*>(vt[buf] + 4 * vr);
                l_run += e;
                a0 += e * vv.x; a1 += e * vv.y; a2 += e * vv.z;
            }
        }
        // ---- column statistics (lane = vertex v0 + l31, lane local over the wavefront's 32 pixels = 16 registers x 2 halves)
        //      and the 2x2 pooling (four consecutive registers)
        float cm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float s = ((mbits >> r) & 1u) ? MASKED_SCORE : accP[r];
            accP[r] = s;
            cm = fmaxf(cm, a.tau_mesh * s);
        }
        cm = fmaxf(cm, other_half(cm));
        float se = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float2 gxy = *reinterpret_cast<const float2*>(&pixgrid[wave][acc_row(r, half)][0]);
            const float e = __expf(a.tau_mesh * accP[r] - cm);
            se += e; sx += e * gxy.x; sy += e * gxy.y;
        }
        se += other_half(se); sx += other_half(sx); sy += other_half(sy);
        if (half == 0) *reinterpret_cast<float4*>(&colred[buf][wave][l31][0]) = make_float4(cm, se, sx, sy);
        float pq[4];
#pragma unroll
        // registers 4j .. 4j+3 = (row 0, x), (row 1, x), (row 0, x+1), (row 1, x+1) of the cell (strip_pixel); summed in the association of
        // ATen's bilinear kernel, horizontal neighbours first -- 0.25 * ((a + b) + (c + d)) -- because at the mask border a cell mixes
        // -1e5 with real scores and the order of the additions decides the rounding (pretrained_corr.py:120-123; G4 fixture)
        for (int j = 0; j < 4; j++) pq[j] = 0.25f * ((accP[4 * j] + accP[4 * j + 2]) + (accP[4 * j + 1] + accP[4 * j + 3]));
        if (v0 + l31 < a.V) {
#pragma unroll
            for (int j = 0; j < 4; j++) pooled_col[(size_t)(2 * j) * a.V + v0] = pq[j];
        }
        if (bridge) {        // workgroup-uniform
            float cm2 = fmaxf(fmaxf(a.tau_mesh * pq[0], a.tau_mesh * pq[1]), fmaxf(a.tau_mesh * pq[2], a.tau_mesh * pq[3]));
            cm2 = fmaxf(cm2, other_half(cm2));
            float se2 = 0.f, sx2 = 0.f, sy2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float e = __expf(a.tau_mesh * pq[j] - cm2);
                se2 += e; sx2 += e * ghx[j]; sy2 += e * ghy[j];
            }
            se2 += other_half(se2); sx2 += other_half(sx2); sy2 += other_half(sy2);
            if (half == 0) *reinterpret_cast<float4*>(&colred2[buf][wave][l31][0]) = make_float4(cm2, se2, sx2, sy2);
        }
        __syncthreads();      // colred[buf] complete; next tile staged; everyone done reading mt[buf]
        if (bridge && tid >= 32 && tid < 64 && v0 + tid - 32 < a.V) {     // lanes 32..63 of wavefront 0: the pooled scores' partials
            const int vq = tid - 32;
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; w++) M = fmaxf(M, colred2[buf][w][vq][0]);
            float se4 = 0.f, sx4 = 0.f, sy4 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float4 c4 = *reinterpret_cast<const float4*>(&colred2[buf][w][vq][0]);
                const float sc = __expf(c4.x - M);
                se4 += c4.y * sc; sx4 += c4.z * sc; sy4 += c4.w * sc;
            }
            *reinterpret_cast<float4*>(a.colpart2 + (((size_t)b * a.nblk + blk) * a.V + v0 + vq) * 4) = make_float4(M, se4, sx4, sy4);
        }
        if (tid < 32 && v0 + tid < a.V) {        // merge the four wavefronts' partials of vertex tid (fixed order)
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; w++) M = fmaxf(M, colred[buf][w][tid][0]);
            float se4 = 0.f, sx4 = 0.f, sy4 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const float4 c4 = *reinterpret_cast<const float4*>(&colred[buf][w][tid][0]);
                const float sc = __expf(c4.x - M);
                se4 += c4.y * sc; sx4 += c4.z * sc; sy4 += c4.w * sc;
            }
            *reinterpret_cast<float4*>(a.colpart + (((size_t)b * a.nblk + blk) * a.V + v0 + tid) * 4) = make_float4(M, se4, sx4, sy4);
        }
    }
    // ---- merge the two halves (disjoint vertex sets of the same pixel) and write match + row statistics
    const float m_o = other_half(m_run), l_o = other_half(l_run), a0_o = other_half(a0), a1_o = other_half(a1), a2_o = other_half(a2);
    if (half == 0) {
        const float M = fmaxf(m_run, m_o);
        const float s1 = __expf(m_run - M), s2 = __expf(m_o - M);
        const float L = l_run * s1 + l_o * s2;
        float* mp = a.match + ((size_t)b * a.P + p) * 3;
        mp[0] = (a0 * s1 + a0_o * s2) / L;
        mp[1] = (a1 * s1 + a1_o * s2) / L;
        mp[2] = (a2 * s1 + a2_o * s2) / L;
        a.rowstat[((size_t)b * a.P + p) * 2 + 0] = M;
        a.rowstat[((size_t)b * a.P + p) * 2 + 1] = L;
    }
}

// column partials of all strips -> imatch[B,2,V], colstat[B,V,2] = (max, sum)   (fixed merge order)
__global__ void fvm_col_merge_kernel(const float* __restrict__ colpart, int B, int nblk, int V, float* __restrict__ imatch,
                                     float* __restrict__ colstat) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V, v = i - b * V;
    const float* cp = colpart + ((size_t)b * nblk * V + v) * 4;
    float M = -INFINITY;
    for (int k = 0; k < nblk; k++) M = fmaxf(M, cp[(size_t)k * V * 4]);
    float se = 0.f, sx = 0.f, sy = 0.f;
    for (int k = 0; k < nblk; k++) {
        const float4 c4 = *reinterpret_cast<const float4*>(cp + (size_t)k * V * 4);
        const float sc = __expf(c4.x - M);
        se += c4.y * sc; sx += c4.z * sc; sy += c4.w * sc;
    }
    imatch[((size_t)b * 2 + 0) * V + v] = sx / se;
    imatch[((size_t)b * 2 + 1) * V + v] = sy / se;
    colstat[((size_t)b * V + v) * 2 + 0] = M;
    colstat[((size_t)b * V + v) * 2 + 1] = se;
}

// ------------------------------------------------------------------------------------------------------------------
// backward A: lane = pixel, contraction over vertices -> g_img[B,64,P]
// ------------------------------------------------------------------------------------------------------------------
#ifndef FVM_IMG_WAVES
#define FVM_IMG_WAVES 2
#endif
#ifndef FVM_MESH_WAVES
#define FVM_MESH_WAVES 3
#endif
constexpr int GPROW = 40;    // LDS row stride of a wavefront's [8 pool cells][32 vertices] gradient tile: the b128 reads of the 8 cells x 2
                             // halves fall on 16 distinct 4-bank groups

__global__ __launch_bounds__(256, FVM_IMG_WAVES) void fvm_backward_img_kernel(const FvmArgs a) {
    __shared__ __attribute__((aligned(16))) float mt[2][32 * MROW];
    __shared__ __attribute__((aligned(16))) float vt[2][32 * 4];
    __shared__ __attribute__((aligned(16))) float ct[2][32 * 8];     // per vertex: cmax, 1/csum, gi0, gi1, d_c
    __shared__ __attribute__((aligned(16))) float gpt[2][4][8 * GPROW];   // per wavefront: g_pooled of its 8 pool cells x the tile's vertices

    int blk, b;
    if (!xcd_unit(a.nblk, a.B, blk, b)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int p = strip_pixel(blk, wave, l31);
    const bool masked = !(a.mask[(size_t)b * a.P + p] > 0.f);
    const float gx = a.grid[p], gy = a.grid[a.P + p];
    float breg[32];
    load_pixel_operand(a, b, p, half, breg);
    const float rmax = a.rowstat_in[((size_t)b * a.P + p) * 2], rinv = 1.f / a.rowstat_in[((size_t)b * a.P + p) * 2 + 1];
    float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f, d_r = 0.f;
    if (a.g_match) {
        const float* gp = a.g_match + ((size_t)b * a.P + p) * 3;
        const float* mp = a.match_in + ((size_t)b * a.P + p) * 3;
        gm0 = gp[0]; gm1 = gp[1]; gm2 = gp[2];
        d_r = gm0 * mp[0] + gm1 * mp[1] + gm2 * mp[2];
    }
    // g_pooled[cell][v]: a lane needs the row of its pixel's 2x2 cell at the tile's vertices -- 16 scattered dwords per tile if read
    // directly (16 cache lines per instruction).  Instead the wavefront loads its 8 cells x 32 vertices with 4 coalesced dword loads
    // (element e = lane + 64 i: cell e >> 5, vertex e & 31), one tile ahead, and hands them over through LDS.
    const bool with_pool = a.g_pooled != nullptr;
    const float* gpool_wave = with_pool ? a.g_pooled + ((size_t)b * (a.P / 4) + blk * (WF / 2) + 8 * wave) * a.V : nullptr;
    auto load_gpool = [&](int v0, float* q) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = lane + 64 * i, vv = v0 + (e & 31);
            q[i] = (with_pool && vv < a.V) ? gpool_wave[(size_t)(e >> 5) * a.V + vv] : 0.f;
        }
    };
    auto store_gpool = [&](const float* q, float* dst) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = lane + 64 * i;
            dst[(e >> 5) * GPROW + (e & 31)] = q[i];
        }
    };

    auto stage_cols = [&](int v0, float* dst) {
        if (tid < 32) {
            float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = make_float4(0.f, 0.f, 0.f, 0.f);
            const int v = v0 + tid;
            if (v < a.V) {
                const float cm = a.colstat[((size_t)b * a.V + v) * 2], cs = a.colstat[((size_t)b * a.V + v) * 2 + 1];
                float g0 = 0.f, g1 = 0.f, dc = 0.f;
                if (a.g_imatch) {
                    g0 = a.g_imatch[((size_t)b * 2 + 0) * a.V + v];
                    g1 = a.g_imatch[((size_t)b * 2 + 1) * a.V + v];
                    dc = g0 * a.imatch[((size_t)b * 2 + 0) * a.V + v] + g1 * a.imatch[((size_t)b * 2 + 1) * a.V + v];
                }
                x0 = make_float4(cm, 1.f / cs, g0, g1);
                x1 = make_float4(dc, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(dst + 8 * tid) = x0;
            *reinterpret_cast<float4*>(dst + 8 * tid + 4) = x1;
        }
    };

    f32x16 g_lo, g_hi;
#pragma unroll
    for (int r = 0; r < 16; r++) { g_lo[r] = 0.f; g_hi[r] = 0.f; }

    stage_vertex_tile(a, b, 0, mt[0], vt[0]);
    stage_cols(0, ct[0]);
    {
        float q[4];
        load_gpool(0, q);
        store_gpool(q, gpt[0][wave]);
    }
    __syncthreads();
    for (int tv = 0; tv < a.ntile; tv++) {
        const int buf = tv & 1, v0 = 32 * tv;
        float gq[4];
        if (tv + 1 < a.ntile) {
            load_gpool(v0 + 32, gq);
            stage_vertex_tile(a, b, v0 + 32, mt[buf ^ 1], vt[buf ^ 1]);
            stage_cols(v0 + 32, ct[buf ^ 1]);
        }
        f32x16 acc = score_tile(mt[buf], breg, l31, half);
        // ---- d S for the lane's pixel and the tile's vertices (zero on masked pixels and padding vertices)
        const float* gprow = gpt[buf][wave] + (l31 >> 2) * GPROW + 4 * half;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 gp4 = *reinterpret_cast<const float4*>(gprow + 8 * j);       // vertices acc_row(4j .. 4j+3, half)
            const float gpv[4] = {gp4.x, gp4.y, gp4.z, gp4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 4 * j + i;
                const int vr = acc_row(r, half);
                float ds = 0.f;
                if (!masked && v0 + vr < a.V) {
                    const float s = acc[r];
                    const float4 vv = *reinterpret_cast<const float4*>(vt[buf] + 4 * vr);
                    const float4 c0 = *reinterpret_cast<const float4*>(ct[buf] + 8 * vr);
                    const float dc = ct[buf][8 * vr + 4];
                    const float pr = __expf(a.tau_img * s - rmax) * rinv;
                    const float pc = __expf(a.tau_mesh * s - c0.x) * c0.y;
                    ds = a.tau_img * pr * (gm0 * vv.x + gm1 * vv.y + gm2 * vv.z - d_r) + a.tau_mesh * pc * (c0.z * gx + c0.w * gy - dc);
                    if (with_pool) ds += 0.25f * gpv[i];
                }
                acc[r] = ds;
            }
        }
        if (tv + 1 < a.ntile) store_gpool(gq, gpt[buf ^ 1][wave]);
        // ---- g_img[c][p] += sum_v mesh[v][c] dS[v][p]: dS registers are the B operand; k-step r pairs the vertices
        //      acc_row(r, 0) and acc_row(r, 1), the A operand is mesh[acc_row(r, half)][c = l31 (+32)]
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float* mrow = mt[buf] + acc_row(r, half) * MROW + l31;
            g_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(mrow[0], acc[r], g_lo, 0, 0, 0);
            g_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(mrow[32], acc[r], g_hi, 0, 0, 0);
        }
        __syncthreads();
    }
    float* gp = a.g_img + (size_t)b * C * a.P + p;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int c = acc_row(r, half);
        gp[(size_t)c * a.P] = g_lo[r];
        gp[(size_t)(c + 32) * a.P] = g_hi[r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward B: lane = vertex, contraction over pixels -> g_mesh[B,V,64]
// ------------------------------------------------------------------------------------------------------------------
constexpr int IROW = 33;     // LDS row stride of an [64 channels][32 pixels] tile: conflict-free along channels and along pixels

__global__ __launch_bounds__(256, FVM_MESH_WAVES) void fvm_backward_mesh_kernel(const FvmArgs a) {
    static_assert(C * IROW >= 32 * 65, "the final cross-wavefront sum reuses the image tiles");
    __shared__ __attribute__((aligned(16))) float it[4][C * IROW];      // per wavefront: image tile; at the end: the wavefront's partial sums
    __shared__ __attribute__((aligned(16))) float pt[4][32 * 12];       // per wavefront, per pixel: rmax, 1/rsum, gm0..2, d_r, gx, gy, live, p2
    float (*red)[C * IROW] = it;

    int tv, b;
    if (!xcd_unit(a.ntile, a.B, tv, b)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int v = 32 * tv + l31;
    const bool vok = v < a.V;
    // B operand of the score tile: the lane's vertex features, breg[4t + r] = mesh[v][8t + 4 half + r]
    float breg[32];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vok) m = *reinterpret_cast<const float4*>(a.mesh + ((size_t)b * a.V + v) * C + 8 * t + 4 * half);
        breg[4 * t + 0] = m.x; breg[4 * t + 1] = m.y; breg[4 * t + 2] = m.z; breg[4 * t + 3] = m.w;
    }
    float vx = 0.f, vy = 0.f, vz = 0.f, cmax = 0.f, cinv = 0.f, g0 = 0.f, g1 = 0.f, d_c = 0.f;
    if (vok) {
        const float* vp = a.verts + ((size_t)b * a.V + v) * 3;
        vx = vp[0]; vy = vp[1]; vz = vp[2];
        cmax = a.colstat[((size_t)b * a.V + v) * 2];
        cinv = 1.f / a.colstat[((size_t)b * a.V + v) * 2 + 1];
        if (a.g_imatch) {
            g0 = a.g_imatch[((size_t)b * 2 + 0) * a.V + v];
            g1 = a.g_imatch[((size_t)b * 2 + 1) * a.V + v];
            d_c = g0 * a.imatch[((size_t)b * 2 + 0) * a.V + v] + g1 * a.imatch[((size_t)b * 2 + 1) * a.V + v];
        }
    }
    f32x16 g_lo, g_hi;
#pragma unroll
    for (int r = 0; r < 16; r++) { g_lo[r] = 0.f; g_hi[r] = 0.f; }

    float* my_it = it[wave];
    float* my_pt = pt[wave];
    const int ntp = a.P / 32;
    for (int tp = wave; tp < ntp; tp += 4) {
        const int p0 = 32 * tp;
        // ---- stage 32 pixels (consecutive in one image row): features [c][p] and the per-pixel scalars (wavefront private)
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int e = lane + 64 * i;               // c = e >> 5, pixel e & 31
            my_it[(e >> 5) * IROW + (e & 31)] = a.img[((size_t)b * C + (e >> 5)) * a.P + p0 + (e & 31)];
        }
        if (lane < 32) {
            const int p = p0 + lane;
            const bool live = a.mask[(size_t)b * a.P + p] > 0.f;
            float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f, dr = 0.f;
            if (a.g_match) {
                const float* gp = a.g_match + ((size_t)b * a.P + p) * 3;
                const float* mp = a.match_in + ((size_t)b * a.P + p) * 3;
                gm0 = gp[0]; gm1 = gp[1]; gm2 = gp[2];
                dr = gm0 * mp[0] + gm1 * mp[1] + gm2 * mp[2];
            }
            const int row = p / WF, x = p - row * WF;
            float* o = my_pt + 12 * lane;
            o[0] = a.rowstat_in[((size_t)b * a.P + p) * 2];
            o[1] = 1.f / a.rowstat_in[((size_t)b * a.P + p) * 2 + 1];
            o[2] = gm0; o[3] = gm1; o[4] = gm2; o[5] = dr;
            o[6] = a.grid[p]; o[7] = a.grid[a.P + p];
            o[8] = live ? 1.f : 0.f;
            o[9] = __int_as_float((row >> 1) * (WF / 2) + (x >> 1));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- S tile: rows = pixels (A operand img[c][p = l31]), columns = vertices (B = breg): acc[r] = S[p = acc_row(r,half)][v]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(my_it[(8 * t + 4 * half + r) * IROW + l31], breg[4 * t + r], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pr_ = acc_row(r, half);
            const float* o = my_pt + 12 * pr_;
            const float4 o0 = *reinterpret_cast<const float4*>(o);
            const float4 o1 = *reinterpret_cast<const float4*>(o + 4);
            const float4 o2 = *reinterpret_cast<const float4*>(o + 8);
            float ds = 0.f;
            if (vok && o2.x > 0.f) {
                const float s = acc[r];
                const float pr = __expf(a.tau_img * s - o0.x) * o0.y;
                const float pc = __expf(a.tau_mesh * s - cmax) * cinv;
                ds = a.tau_img * pr * (o0.z * vx + o0.w * vy + o1.x * vz - o1.y) + a.tau_mesh * pc * (g0 * o1.z + g1 * o1.w - d_c);
                if (a.g_pooled) ds += 0.25f * a.g_pooled[((size_t)b * (a.P / 4) + __float_as_int(o2.y)) * a.V + v];
            }
            acc[r] = ds;
        }
        // ---- g_mesh^T[c][v] += sum_p img[c][p] dS[p][v]
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int pr_ = acc_row(r, half);
            g_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(my_it[l31 * IROW + pr_], acc[r], g_lo, 0, 0, 0);
            g_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(my_it[(l31 + 32) * IROW + pr_], acc[r], g_hi, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- sum the four wavefronts (fixed order) and store g_mesh[b][v][c]; lane holds [c = acc_row(r,half) (+32)][v = l31]
    //      (red[wave] is the wavefront's own image tile, which only it reads: no barrier needed before overwriting it)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        red[wave][l31 * 65 + acc_row(r, half)] = g_lo[r];
        red[wave][l31 * 65 + acc_row(r, half) + 32] = g_hi[r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * C; e += 256) {
        const int vl = e >> 6, c = e & 63;
        if (32 * tv + vl < a.V)
            a.g_mesh[((size_t)b * a.V + 32 * tv + vl) * C + c] =
                (red[0][vl * 65 + c] + red[1][vl * 65 + c]) + (red[2][vl * 65 + c] + red[3][vl * 65 + c]);
    }
}

int check_shape(int B, int Cf, int hf, int wf, int V) {
    if (B <= 0 || V <= 0) return scp::fail(hipErrorInvalidValue, "feature_vertex_match: empty problem");
    if (Cf != C || wf != WF || hf < 2 || (hf & 1))
        return scp::fail(hipErrorInvalidValue, "feature_vertex_match (fused): needs 64 channels, a 64-pixel-wide map and an even height");
    return 0;
}

}  // namespace

extern "C" size_t scp_fvm_workspace(int B, int hf, int V) { return 2 * (size_t)B * (hf / 2) * V * 4 * sizeof(float); }

extern "C" int scp_fvm_forward(const float* img_feat, const float* mesh_feat, const float* mask_down, const float* verts,
                               const float* grid, float tau_img, float tau_mesh, int B, int Cf, int hf, int wf, int V,
                               float* pooled, float* match, float* imatch, float* rowstat, float* colstat, const float* grid_half,
                               float* bridge_xy, float* bridge_colstat, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_shape(B, Cf, hf, wf, V)) return e;
    if (workspace_bytes < scp_fvm_workspace(B, hf, V)) return scp::fail(hipErrorInvalidValue, "feature_vertex_match: workspace too small");
    FvmArgs a{};
    a.img = img_feat; a.mesh = mesh_feat; a.mask = mask_down; a.verts = verts; a.grid = grid;
    a.tau_img = tau_img; a.tau_mesh = tau_mesh;
    a.B = B; a.P = hf * wf; a.V = V; a.hf = hf; a.ntile = (V + 31) / 32; a.nblk = hf / 2;
    if ((grid_half != nullptr) != (bridge_xy != nullptr) || (grid_half != nullptr) != (bridge_colstat != nullptr))
        return scp::fail(hipErrorInvalidValue, "feature_vertex_match: grid_half, bridge_xy and bridge_colstat come together");
    a.pooled = pooled; a.match = match; a.rowstat = rowstat; a.colpart = static_cast<float*>(workspace);
    a.grid_half = grid_half; a.colpart2 = a.colpart + (size_t)B * a.nblk * V * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(fvm_forward_kernel, dim3(xcd_grid(a.nblk, B)), dim3(256), 0, st, a);
    if (int e = scp::check_launch("fvm_forward")) return e;
    hipLaunchKernelGGL(fvm_col_merge_kernel, dim3((B * V + 255) / 256), dim3(256), 0, st, a.colpart, B, a.nblk, V, imatch, colstat);
    if (int e = scp::check_launch("fvm_col_merge")) return e;
    if (grid_half) {
        hipLaunchKernelGGL(fvm_col_merge_kernel, dim3((B * V + 255) / 256), dim3(256), 0, st, a.colpart2, B, a.nblk, V, bridge_xy, bridge_colstat);
        return scp::check_launch("fvm_col_merge (pooled)");
    }
    return 0;
}

extern "C" int scp_fvm_backward(const float* img_feat, const float* mesh_feat, const float* mask_down, const float* verts,
                                const float* grid, float tau_img, float tau_mesh, int B, int Cf, int hf, int wf, int V,
                                const float* match, const float* imatch, const float* rowstat, const float* colstat,
                                const float* g_match, const float* g_imatch, const float* g_pooled, float* g_img_feat,
                                float* g_mesh_feat, void* stream) {
    if (int e = check_shape(B, Cf, hf, wf, V)) return e;
    FvmArgs a{};
    a.img = img_feat; a.mesh = mesh_feat; a.mask = mask_down; a.verts = verts; a.grid = grid;
    a.tau_img = tau_img; a.tau_mesh = tau_mesh;
    a.B = B; a.P = hf * wf; a.V = V; a.hf = hf; a.ntile = (V + 31) / 32; a.nblk = hf / 2;
    a.match_in = match; a.imatch = imatch; a.rowstat_in = rowstat; a.colstat = colstat;
    a.g_match = g_match; a.g_imatch = g_imatch; a.g_pooled = g_pooled; a.g_img = g_img_feat; a.g_mesh = g_mesh_feat;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_img_feat) {
        hipLaunchKernelGGL(fvm_backward_img_kernel, dim3(xcd_grid(a.nblk, B)), dim3(256), 0, st, a);
        if (int e = scp::check_launch("fvm_backward_img")) return e;
    }
    if (g_mesh_feat) {
        hipLaunchKernelGGL(fvm_backward_mesh_kernel, dim3(xcd_grid(a.ntile, B)), dim3(256), 0, st, a);
        if (int e = scp::check_launch("fvm_backward_mesh")) return e;
    }
    return 0;
}
