// self-corr-pose_amd/csrc/conv_stem.hip -- the ResNet stem: 7x7 / stride 2 / pad 3 convolution of the 3-channel image, 3 -> 64
// channels, forward (+ the batch statistics of the BatchNorm that follows) and weight gradient, on the gfx950 fp32 matrix cores.
//
// Replaces MIOpen for model/module/network/image_encoder.py:122-124 (torchvision ResNet18 conv1 + bn1 as the image encoder's
// first layer), called twice per step (encoder.py:29-37, correspondence.py:91).  The image carries no gradient: no input gradient.
//
// Why its own kernel: with Cin = 3 a tap is 12 bytes -- nothing for the 16-byte LDS-DMA rows of csrc/conv_igemm.hip to move, and
// K = 3 x 7 x 7 = 147 is too short for a staged main loop.  Instead a workgroup takes 128 consecutive output pixels of one output
// row, loads the 3 x 7 input rows they touch ONCE into LDS (NCHW input: a row is contiguous; 21 x 261 floats, zero-filled outside
// the image) and every matrix-core operand is a single conflict-free ds_read_b32 of that block:
//     forward : y[pixel][co]  = sum_k  halo[row(k)][2 pixel + kx(k)] * w[co][k]          A = halo (row = pixel), B = weights
//     wgrad   : dw[co][k]     = sum_pixel dy[pixel][co] * halo[row(k)][2 pixel + kx(k)]   A = dy (row = co),     B = halo (col = k)
// with v_mfma_f32_32x32x2_f32 (exact fp32 products -- this layer does not use the split-bf16 loop: 2 x 9.9 GFLOP per pass,
// 2 % of the encoder's work).  The two k of one MFMA step belong to the two lane halves, and a lane's LDS address must be
// "lane base + compile-time offset": the 147 k are therefore PAIRED so that the partner of k sits at a fixed distance --
// (kx, kx + 1) for kx = 0, 2, 4 of each of the 21 (ci, ky) rows (distance 1), and the kx = 6 taps of rows (2t, 2t + 1) (distance
// one row); 63 + 11 = 74 steps, one padding k with zero weight.
// Forward epilogue: raw NHWC output + per-tile column sums and sums of squares; the last workgroup folds them in fp64 in tile
// order and finalises the BatchNorm statistics (csrc/bn_common.h) -- the protocol of csrc/conv_igemm.hip.
// Weight gradient: persistent workgroups keep their 64 x 160 accumulator block in registers over all their tiles and write one
// partial block each; a second kernel adds the partial blocks in workgroup order (deterministic, no atomics).
// Roofline: bound = fp32 MFMA (2 x 148 x 64 flop per pixel at 256 flop / cycle / CU); algorithmic bytes 4 (3 H W + 64 Ho Wo) per image.
#include <hip/hip_runtime.h>

#include "bn_common.h"
#include "gemm_core.h"
#include "scp_common.h"
#include "scp_hip.h"

namespace {

using scp::f32x16;

constexpr int THREADS = 256, TILE = 128, CO = 64, KROWS = 21, HW = 264, NSTEP = 74;
constexpr int HALO_FLOATS = (KROWS + 1) * HW;            // + one row that only the zero-weight padding k reads
constexpr int WL_STRIDE = CO + 1;                        // weights in LDS: [2 NSTEP][CO + 1]

struct StemArgs {
    const float* x;         // [N, 3, H, W]
    const float* w;         // [64, 3, 7, 7] with element strides ws_*
    long ws_co, ws_ci, ws_ky, ws_kx;
    float* y;               // forward: [N, Ho, Wo, 64] raw output
    const float* dy;        // wgrad: [N, Ho, Wo, 64]
    float* partials;        // forward: [2][tiles][64]; wgrad: [workgroups][64][160]
    unsigned* ticket;
    scp_bn::FwdFinalize fin;
    int N, H, W, Ho, Wo, tiles_x, tiles;
};

// step s, lane half h -> (row = ci * 7 + ky, kx) of its k; row == KROWS: the padding k
__host__ __device__ constexpr int step_row(int s, int h) { return s < 63 ? s / 3 : 2 * (s - 63) + h; }
__host__ __device__ constexpr int step_kx(int s, int h) { return s < 63 ? 2 * (s % 3) + h : 6; }
// halo offset of (step s, half 0); half 1 adds 1 (s < 63) or HW (s >= 63)
__host__ __device__ constexpr int step_off(int s) { return step_row(s, 0) * HW + step_kx(s, 0); }

// the 21 input rows x 264 columns [2 x0 - 4, 2 x0 + 260) the tile (img, yo, x0 .. x0 + 127) touches, as 21 x 66 float4 (x0 is a
// multiple of 128 and W of 4: a float4 is inside the image or outside as a whole; outside -> zeros).  fetch() leaves them in
// registers -- issued before the previous tile's MFMAs, they are in flight behind them --, store() puts them into the LDS block.
constexpr int HALO_V4 = KROWS * (HW / 4), HALO_PER = (HALO_V4 + THREADS - 1) / THREADS;
struct HaloLoader {
    float4 v[HALO_PER];
    __device__ __forceinline__ void fetch(const StemArgs& g, int img, int yo, int x0) {
#pragma unroll
        for (int i = 0; i < HALO_PER; i++) {
            const int idx = threadIdx.x + THREADS * i;
            const int row = idx / (HW / 4), q = idx - row * (HW / 4);
            const int ci = row / 7, ky = row - ci * 7;
            const int yy = 2 * yo - 3 + ky, xx = 2 * x0 - 4 + 4 * q;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < HALO_V4 && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W)
                v[i] = *reinterpret_cast<const float4*>(g.x + ((size_t)(img * 3 + ci) * g.H + yy) * g.W + xx);
        }
    }
    __device__ __forceinline__ void store(float* halo) const {
#pragma unroll
        for (int i = 0; i < HALO_PER; i++) {
            const int idx = threadIdx.x + THREADS * i;
            if (idx < HALO_V4) reinterpret_cast<float4*>(halo)[idx] = v[i];
        }
    }
};

__device__ __forceinline__ void tile_coords(const StemArgs& g, int tile, int& img, int& yo, int& x0) {
    const int per_img = g.Ho * g.tiles_x;
    img = tile / per_img;
    const int rem = tile - img * per_img;
    yo = rem / g.tiles_x;
    x0 = (rem - yo * g.tiles_x) * TILE;
}

__global__ __launch_bounds__(THREADS, 2) void stem_forward_kernel(const StemArgs g) {
    __shared__ __attribute__((aligned(16))) float halo[HALO_FLOATS];
    __shared__ float wl[2 * NSTEP * WL_STRIDE];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // weights, in step order: wl[2 s + h][co] = w[co][k(s, h)]; 37 elements per thread, their loads issued together
    {
        constexpr int PER = (2 * NSTEP * CO + THREADS - 1) / THREADS;
        float v[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int idx = threadIdx.x + THREADS * i;
            const int s2 = idx % (2 * NSTEP), co = min(idx / (2 * NSTEP), CO - 1);
            const int s = s2 >> 1, h = s2 & 1;
            const int row = step_row(s, h), kx = step_kx(s, h);
            const int rowc = min(row, KROWS - 1), ci = rowc / 7, ky = rowc - ci * 7;
            const float t = g.w[co * g.ws_co + ci * g.ws_ci + ky * g.ws_ky + kx * g.ws_kx];
            v[i] = row < KROWS ? t : 0.f;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int idx = threadIdx.x + THREADS * i;
            if (idx < 2 * NSTEP * CO) wl[(idx % (2 * NSTEP)) * WL_STRIDE + idx / (2 * NSTEP)] = v[i];
        }
    }
    for (int idx = KROWS * HW + threadIdx.x; idx < HALO_FLOATS; idx += THREADS) halo[idx] = 0.f;
    // halo column of (pixel p, tap kx) = 2 p + kx + 1 (the block starts one column left of the first tap: float4 alignment)
    const float* a_lo = halo + 2 * (32 * wave + l31) + 1 + half;          // steps < 63
    const float* a_hi = halo + 2 * (32 * wave + l31) + 1 + half * HW;     // steps >= 63
    const float* b_rd = wl + half * WL_STRIDE + l31;

    HaloLoader ld;
    int img, yo, x0;
    if ((int)blockIdx.x < g.tiles) {
        tile_coords(g, blockIdx.x, img, yo, x0);
        ld.fetch(g, img, yo, x0);
    }
    for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
        tile_coords(g, tile, img, yo, x0);
        __syncthreads();                                 // the previous tile's reads of the halo block (and the weight fill) are done
        ld.store(halo);
        __syncthreads();
        if (tile + (int)gridDim.x < g.tiles) {
            int img2, yo2, x02;
            tile_coords(g, tile + gridDim.x, img2, yo2, x02);
            ld.fetch(g, img2, yo2, x02);
        }
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][r] = acc[1][r] = 0.f;
        scp::static_for<0, NSTEP>([&](auto ss) {
            constexpr int s = decltype(ss)::value;
            const float a = s < 63 ? a_lo[step_off(s)] : a_hi[step_off(s)];
            const float b0 = b_rd[2 * s * WL_STRIDE], b1 = b_rd[2 * s * WL_STRIDE + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        });
        // raw output + the tile's column sums (rows beyond the image row are not stored and not counted)
        float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
        const size_t row0 = ((size_t)(img * g.Ho + yo)) * g.Wo;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int xo = x0 + 32 * wave + scp::acc_row(r, half);
            if (xo < g.Wo) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float v = acc[j][r];
                    csum[j] += v;
                    csq[j] += v * v;
                    g.y[(row0 + xo) * CO + 32 * j + l31] = v;
                }
            }
        }
        if (g.partials) {
            __syncthreads();                             // every wavefront has finished reading the halo block: it holds the fold now
            float* red = halo;                           // [4 waves][2 j][2 which][32]
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float s = csum[j] + __shfl_xor(csum[j], 32), q = csq[j] + __shfl_xor(csq[j], 32);
                if (half == 0) {
                    red[((wave * 2 + j) * 2 + 0) * 32 + l31] = s;
                    red[((wave * 2 + j) * 2 + 1) * 32 + l31] = q;
                }
            }
            __syncthreads();
            if (threadIdx.x < 2 * CO) {
                const int which = threadIdx.x / CO, col = threadIdx.x - which * CO;
                const int j = col >> 5, l = col & 31;
                float s = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; wv++) s += red[((wv * 2 + j) * 2 + which) * 32 + l];
                __hip_atomic_store(g.partials + ((size_t)which * g.tiles + tile) * CO + col, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (g.partials && g.ticket) {
        __syncthreads();
        if (scp_bn::last_block_arrived(g.ticket, reinterpret_cast<int*>(halo))) {
            if (threadIdx.x == 0 && g.fin.batches_tracked) *g.fin.batches_tracked += 1;
            const int tc_n = CO / 4;
            double sa[4], sb[4];
            __syncthreads();
            if (!scp_bn::fold_partials(g.partials, g.partials + (size_t)g.tiles * CO, g.tiles, CO, tc_n, sa, sb, reinterpret_cast<float4*>(halo)))
                return;
            const int tc = threadIdx.x % tc_n;
#pragma unroll
            for (int i = 0; i < 4; i++) scp_bn::finalize_channel(g.fin, 4 * tc + i, true, sa[i], sb[i]);
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
// wavefront = (co tile wave & 1, pixel half wave >> 1): 32 co x 160 k-columns (five accumulator tiles) over the 64 pixels of its
// half of every tile the workgroup takes.  Column n of the K axis: n = row * 7 + kx (= the weight's own (ci, ky, kx) order).
constexpr int NCOL = 160, NT = NCOL / 32;

__global__ __launch_bounds__(THREADS, 2) void stem_wgrad_kernel(const StemArgs g) {
    __shared__ __attribute__((aligned(16))) float halo[HALO_FLOATS];
    __shared__ __attribute__((aligned(16))) float dyl[TILE * CO];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int cot = wave & 1, ph = wave >> 1;
    for (int idx = KROWS * HW + threadIdx.x; idx < HALO_FLOATS; idx += THREADS) halo[idx] = 0.f;
    // B operand: halo[row(n)][2 pixel + kx(n)], pixel = 64 ph + 2 step + half
    const float* b_rd[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = min(32 * t + l31, 146);            // columns 147 .. 159 do not exist: their accumulators are never stored
        const int row = n / 7, kx = n - row * 7;
        b_rd[t] = halo + row * HW + kx + 1 + 2 * (64 * ph + half);
    }
    const float* a_rd = dyl + (64 * ph + half) * CO + 32 * cot + l31;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // dy tile: 128 pixels x 64 channels, contiguous in NHWC (2048 float4, 8 per thread); pixels beyond the row are zeros
    constexpr int DY_PER = TILE * CO / 4 / THREADS;
    float4 dv[DY_PER];
    auto fetch_dy = [&](int img, int yo, int x0) {
        const float* src = g.dy + (((size_t)(img * g.Ho + yo)) * g.Wo + x0) * CO;
        const int npix = min(TILE, g.Wo - x0);
#pragma unroll
        for (int i = 0; i < DY_PER; i++) {
            const int idx = threadIdx.x + THREADS * i;
            dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx / (CO / 4) < npix) dv[i] = *reinterpret_cast<const float4*>(src + 4 * (size_t)idx);
        }
    };
    HaloLoader ld;
    int img, yo, x0;
    if ((int)blockIdx.x < g.tiles) {
        tile_coords(g, blockIdx.x, img, yo, x0);
        ld.fetch(g, img, yo, x0);
        fetch_dy(img, yo, x0);
    }
    for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
        __syncthreads();
        ld.store(halo);
#pragma unroll
        for (int i = 0; i < DY_PER; i++) reinterpret_cast<float4*>(dyl)[threadIdx.x + THREADS * i] = dv[i];
        __syncthreads();
        if (tile + (int)gridDim.x < g.tiles) {
            tile_coords(g, tile + gridDim.x, img, yo, x0);
            ld.fetch(g, img, yo, x0);
            fetch_dy(img, yo, x0);
        }
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const float a = a_rd[2 * s * CO];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b_rd[t][4 * s], acc[t], 0, 0, 0);
        }
    }
    // the two pixel halves of a co tile are added through LDS in a fixed order (5 x 16 x 64 floats per co tile: co tile 0 through the
    // dy tile's memory, co tile 1 through the halo block's), then one partial block per workgroup
    __syncthreads();
    float* dst = cot == 0 ? dyl : halo;
    if (ph == 1) {
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) dst[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (ph == 0) {
        float* out = g.partials + (size_t)blockIdx.x * CO * NCOL;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int n = 32 * t + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = 32 * cot + scp::acc_row(r, half);
                out[co * NCOL + n] = acc[t][r] + dst[(t * 16 + r) * 64 + lane];
            }
        }
    }
}

// dw[co][ci][ky][kx] (element strides ds_*) = sum over the workgroups' partial blocks, in workgroup order
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ partial, float* __restrict__ dw, long ds_co, long ds_ci,
                                                              long ds_ky, long ds_kx, int blocks) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;                   // (co, n): co = i / 147, n = i % 147
    float s = 0.f;
    if (i < CO * 147) {
        const int co = i / 147, n = i - co * 147;
        const float* p = partial + co * NCOL + n;
        for (int k = j; k < blocks; k += 16) s += p[(size_t)k * CO * NCOL];
    }
    red[j][o] = s;
    __syncthreads();
    if (j == 0 && i < CO * 147) {
        float t = red[0][o];
#pragma unroll
        for (int k = 1; k < 16; k++) t += red[k][o];
        const int co = i / 147, n = i - co * 147;
        const int row = n / 7, kx = n - row * 7, ci = row / 7, ky = row - ci * 7;
        dw[co * ds_co + ci * ds_ci + ky * ds_ky + kx * ds_kx] = t;
    }
}

int check_shape(int N, int H, int W, const char* what) {
    if (N <= 0 || H < 8 || W < 8 || (H & 1) || (W & 3)) return scp::fail(hipErrorInvalidValue, what);
    if ((long)N * 3 * H * W >= (1l << 31) || (long)N * (H / 2) * (W / 2) * CO >= (1l << 31)) return scp::fail(hipErrorInvalidValue, what);
    return 0;
}
void set_geometry(StemArgs& g, int N, int H, int W) {
    g.N = N; g.H = H; g.W = W; g.Ho = H / 2; g.Wo = W / 2;
    g.tiles_x = (g.Wo + TILE - 1) / TILE;
    g.tiles = N * g.Ho * g.tiles_x;
}
constexpr int WGRAD_BLOCKS = 512;

}  // namespace

extern "C" int scp_stem_conv_tiles(int N, int H, int W) { return N * (H / 2) * ((W / 2 + TILE - 1) / TILE); }

extern "C" int scp_stem_conv_forward_bn(const float* x, const float* w, long long ws_co, long long ws_ci, long long ws_ky, long long ws_kx, float* y,
                                        int N, int H, int W, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                        long long* batches_tracked, float momentum, float eps, float* save_mean, float* save_invstd,
                                        float* save_scale, float* save_shift, void* workspace, size_t workspace_bytes, unsigned* ticket,
                                        void* stream) {
    if (!x || !w || !y) return scp::fail(hipErrorInvalidValue, "stem_conv_forward: null argument");
    if (int e = check_shape(N, H, W, "stem_conv_forward: needs H even, W a multiple of 4, both >= 8, and tensors below 2^31 elements")) return e;
    StemArgs g{};
    g.x = x; g.w = w; g.ws_co = ws_co; g.ws_ci = ws_ci; g.ws_ky = ws_ky; g.ws_kx = ws_kx; g.y = y;
    set_geometry(g, N, H, W);
    if (workspace) {
        if (!save_mean || !save_invstd || !save_scale || !save_shift || !ticket) return scp::fail(hipErrorInvalidValue, "stem_conv_forward_bn: null argument");
        if (workspace_bytes < (size_t)2 * g.tiles * CO * sizeof(float)) return scp::fail(hipErrorInvalidValue, "stem_conv_forward_bn: workspace too small");
        g.partials = static_cast<float*>(workspace);
        g.ticket = ticket;
        g.fin = scp_bn::FwdFinalize{(long)N * g.Ho * g.Wo, gamma, beta, running_mean, running_var, batches_tracked, momentum, eps,
                                    save_mean, save_invstd, save_scale, save_shift};
    }
    const int grid = g.tiles < 512 ? g.tiles : 512;          // two resident workgroups per CU, each fills its weight block once
    hipLaunchKernelGGL(stem_forward_kernel, dim3(grid), dim3(THREADS), 0, static_cast<hipStream_t>(stream), g);
    return scp::check_launch("stem_conv_forward");
}

extern "C" size_t scp_stem_conv_weight_grad_workspace(int N, int H, int W) {
    (void)N; (void)H; (void)W;
    return (size_t)WGRAD_BLOCKS * CO * NCOL * sizeof(float);
}

extern "C" int scp_stem_conv_weight_grad(const float* x, const float* dy, float* dw, long long ds_co, long long ds_ci, long long ds_ky,
                                         long long ds_kx, void* workspace, size_t workspace_bytes, int N, int H, int W, void* stream) {
    if (!x || !dy || !dw || !workspace) return scp::fail(hipErrorInvalidValue, "stem_conv_weight_grad: null argument");
    if (int e = check_shape(N, H, W, "stem_conv_weight_grad: needs H even, W a multiple of 4, both >= 8, and tensors below 2^31 elements")) return e;
    if (workspace_bytes < scp_stem_conv_weight_grad_workspace(N, H, W)) return scp::fail(hipErrorInvalidValue, "stem_conv_weight_grad: workspace too small");
    StemArgs g{};
    g.x = x; g.dy = dy; g.partials = static_cast<float*>(workspace);
    set_geometry(g, N, H, W);
    const int grid = g.tiles < WGRAD_BLOCKS ? g.tiles : WGRAD_BLOCKS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(THREADS), 0, st, g);
    hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3((CO * 147 + 15) / 16), dim3(256), 0, st, static_cast<const float*>(workspace), dw, (long)ds_co,
                       (long)ds_ci, (long)ds_ky, (long)ds_kx, grid);
    return scp::check_launch("stem_conv_weight_grad");
}
