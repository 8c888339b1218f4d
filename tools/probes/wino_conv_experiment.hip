// tools/probes/wino_conv_experiment.hip -- EXPERIMENT, NOT PART OF THE PRODUCT (not built into libscp_hip.so).
// Round-2 attempt at VERDICT item 8: correct (matched a float64 conv to 2e-6 on all encoder shapes, forward and input
// gradient) but SLOWER than MIOpen's implicit GEMM on MI355X: 142-145 us vs 91 us per layer1/layer2 convolution, 380 us vs
// 92 us for layer4 (64 workgroups only), i.e. 25-73 TFLOP/s direct-equivalent against MIOpen's 100-119.  One wavefront per
// SIMD (256 accumulator registers for the 16 Winograd planes), two barriers per 8-channel stage and register spills leave
// the matrix pipe ~20 % busy; a competitive version needs the planes split over wavefronts (<= 128 accumulators, two
// wavefronts per SIMD) with an LDS exchange before the output transform.  Kept for the record; see DESIGN.md section 7.
//
// 3x3 / stride 1 / pad 1 convolution of the image encoder as a FUSED Winograd
// F(2x2, 3x3) on the gfx950 fp32 matrix cores, NHWC, forward and input-gradient (the same kernel on the flipped filter).
//
// Replaces the MIOpen implicit-GEMM launches behind the nn.Conv2d(.., 3, 1, 1) layers of the reference's encoder
// (model/module/network/image_encoder.py:119-193: torchvision resnet18 BasicBlocks + the U-decoder's conv units), which run
// twice per training step (encoder.py:29-37 and correspondence.py:91).  13 + 6 such layers carry 90 % of the encoder's
// convolution flops.  Winograd needs 16 multiplies per 2x2 output tile and channel pair instead of 36 (2.25x fewer); done
// as separate transform / batched-GEMM / transform kernels the 4x larger intermediates make it HBM-bound in fp32 (measured
// estimate in DESIGN.md), so everything is fused:
//
//   workgroup (4 wavefronts, ONE per SIMD, 512 VGPRs each) = 64 tiles (8 x 8 tiles of one image, or whole small images) x
//   64 output channels.  Per chunk of 8 input channels:
//     * the raw input patch ((2 tbh + 2) x (2 tbw + 2) pixels x 8 channels, zero padded) goes global -> registers -> LDS;
//     * 128 lanes transform it: V = B^T d B per (tile, channel)  -> LDS  V[16 planes][64 tiles][8];
//     * the matching slice of the pre-transformed filter U[16][Cout][Cin] arrives by LDS-DMA -> U[16][64][8];
//     * every wavefront multiplies ALL 16 planes of its 32 tiles x 32 channels: v_mfma_f32_32x32x2_f32, one ds_read_b128
//       per operand and plane feeds 4 MFMAs; 16 planes x 16 accumulator registers = 256 VGPRs.
//   Because one lane holds all 16 planes of its (tile, channel) accumulators, the output transform Y = A^T M A is lane
//   local: no exchange, 2x2 pixels per tile stored straight to NHWC (32 lanes = 32 consecutive channels = 128 B).
//   Loads of chunk k+2, the transform of chunk k+1 and the MFMAs of chunk k overlap (three-deep software pipeline).
//
// dgrad: dx = conv(dy, flip(w)^T) -- same kernel, U built from the flipped / transposed filter (scp_wino_filter_transform).
// Roofline: fp32 MFMA, algorithmic flops = 2 * 16 * tiles * Cin * Cout per launch (the Winograd count; 2.25x below the
// direct 2 * 9 * pixels * Cin * Cout).  Numerics: F(2x2,3x3) transforms are exact in binary (entries 0, +-1, +-1/2); sums
// differ from a direct convolution by fp32 round-off only (tests: vs float64 conv, <= 2e-6 of the output scale).
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int THREADS = 256;
constexpr int TILES = 64;        // tiles per workgroup
constexpr int COB = 64;          // output channels per workgroup
constexpr int KC = 8;            // input channels per pipeline stage
constexpr int MAX_PATCH = 408;   // pixels of the raw patch: 18 x 18 (one 8x8-tile block) or 4 images of 10 x 10

#define SCP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SCP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct WinoArgs {
    const float* x;      // [N, H, W, Cin]
    const float* U;      // [16, Cout, Cin]
    const float* bias;   // [Cout] or null
    float* y;            // [N, H, W, Cout]
    int N, H, W, Cin, Cout;
    int tbh, tbw, nb;            // tile block: nb images x tbh x tbw tiles (= 64)
    int blocks_h, blocks_w;      // tile blocks per image (1 when nb > 1)
    int img_groups;              // ceil(N / nb)
    int co_blocks;
    int ph, pw;                  // raw patch of one image: (2 tbh + 2) x (2 tbw + 2) pixels
    float neg_slope;             // epilogue: y = v >= 0 ? v : neg_slope * v   (1 = identity)
};

// 4x4 input tile (per channel) -> Winograd domain, in place on float4 = 4 channels: V = B^T d B
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(THREADS, 1) void wino_conv_kernel(const WinoArgs g) {
    // separate LDS objects per pipeline stage (the compiler tracks LDS-DMA hazards per object, see vit_gemm.hip)
    __shared__ __attribute__((aligned(16))) float raw0[MAX_PATCH * KC];
    __shared__ __attribute__((aligned(16))) float raw1[MAX_PATCH * KC];
    __shared__ __attribute__((aligned(16))) float v0[16 * TILES * KC];
    __shared__ __attribute__((aligned(16))) float v1[16 * TILES * KC];
    __shared__ __attribute__((aligned(16))) float u0[16 * COB * KC];
    __shared__ __attribute__((aligned(16))) float u1[16 * COB * KC];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tg = wave >> 1, cg = wave & 1;            // wavefront: tiles 32 tg .. +31, channels 32 cg .. +31

    // ---- which tiles / channels.  blockIdx.x = ((img_group * blocks_h + bh) * blocks_w + bw) * co_blocks + cb: the channel
    // blocks of one tile block are adjacent (share the raw patch through L2)
    int b = blockIdx.x;
    const int cb = b % g.co_blocks; b /= g.co_blocks;
    const int bw = b % g.blocks_w; b /= g.blocks_w;
    const int bh = b % g.blocks_h; b /= g.blocks_h;
    const int n0 = b * g.nb;                              // first image of the block
    const int th0 = bh * g.tbh, tw0 = bw * g.tbw;         // first tile (per image)
    const int co0 = cb * COB;
    const int tiles_per_img = g.tbh * g.tbw;
    const int patch_px = g.ph * g.pw;                     // per image
    const int npatch = g.nb * patch_px;

    // ---- raw patch loader: pixel q of the block's patch (q = img * patch_px + py * pw + px), 8 channels = 2 float4 per
    // pixel and stage; 2 * npatch float4 over 256 threads = up to 4 per thread.  Out-of-image pixels are zero.
    constexpr int RAW_PER_THREAD = (2 * MAX_PATCH + THREADS - 1) / THREADS;   // 4
    const float* rsrc[RAW_PER_THREAD];
    bool rok[RAW_PER_THREAD];
#pragma unroll
    for (int i = 0; i < RAW_PER_THREAD; i++) {
        const int e = tid + THREADS * i;                  // float4 slot: pixel e >> 1, channel quad e & 1
        const int q = e >> 1;
        rok[i] = false;
        rsrc[i] = g.x;
        if (q < npatch) {
            const int im = q / patch_px, r = q - im * patch_px;
            const int py = r / g.pw, px = r - py * g.pw;
            const int n = n0 + im, h = 2 * th0 - 1 + py, w = 2 * tw0 - 1 + px;
            if (n < g.N && h >= 0 && h < g.H && w >= 0 && w < g.W) {
                rok[i] = true;
                rsrc[i] = g.x + (((size_t)n * g.H + h) * g.W + w) * g.Cin + 4 * (e & 1);
            }
        }
    }
    float4 rreg[RAW_PER_THREAD];
    auto load_raw = [&](int kc) {
#pragma unroll
        for (int i = 0; i < RAW_PER_THREAD; i++)
            rreg[i] = rok[i] ? *reinterpret_cast<const float4*>(rsrc[i] + kc * KC) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_raw = [&](float* raw) {
#pragma unroll
        for (int i = 0; i < RAW_PER_THREAD; i++) {
            const int e = tid + THREADS * i;
            if (e < 2 * npatch) *reinterpret_cast<float4*>(raw + 4 * e) = rreg[i];
        }
    };

    // ---- filter slice by LDS-DMA: stage = 16 planes x 64 channels x 8 floats = 1024 rows of 32 B = 2048 chunks of 16 B =
    // 32 instructions of 64 lanes; wavefront w issues 8 of them.  Row (plane, co): global U[(plane * Cout + co0 + co) * Cin + kc*8]
    unsigned uoff[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int chunk = (8 * wave + i) * 64 + lane;     // 0..2047 = row * 2 + half-row
        const int row = chunk >> 1, plane = row >> 6, co = row & 63;
        uoff[i] = (unsigned)((plane * g.Cout + min(co0 + co, g.Cout - 1)) * g.Cin + 4 * (chunk & 1));
    }
    auto issue_u = [&](int kc, float* udst) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(g.U + kc * KC + uoff[i]), SCP_LDS_PTR(udst + (8 * wave + i) * 256), 16, 0, 0);
    };

    // ---- input transform: lanes 0..127 own (tile = t_id, channel quad = t_q)
    const int t_id = tid >> 1, t_q = tid & 1;
    int t_base = 0;                                       // float offset of the tile's top-left patch pixel, channel quad
    bool t_live = tid < 2 * TILES;
    {
        const int im = t_id / tiles_per_img, r = t_id - im * tiles_per_img;
        const int ty = r / g.tbw, tx = r - ty * g.tbw;
        t_live = t_live && im < g.nb;                     // tile slots past nb images stay zero (V is cleared once below)
        t_base = ((im * g.ph + 2 * ty) * g.pw + 2 * tx) * KC + 4 * t_q;
    }
    if (g.nb * tiles_per_img < TILES) {
        for (int i = tid; i < 16 * TILES * KC; i += THREADS) { v0[i] = 0.f; v1[i] = 0.f; }
        __syncthreads();
    }
    auto transform = [&](const float* raw, float* vdst) {
        if (!t_live) return;
        float4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {                     // t = B^T d, one patch column at a time
            const float4 d0 = *reinterpret_cast<const float4*>(raw + t_base + (0 * g.pw + j) * KC);
            const float4 d1 = *reinterpret_cast<const float4*>(raw + t_base + (1 * g.pw + j) * KC);
            const float4 d2 = *reinterpret_cast<const float4*>(raw + t_base + (2 * g.pw + j) * KC);
            const float4 d3 = *reinterpret_cast<const float4*>(raw + t_base + (3 * g.pw + j) * KC);
            t[0][j] = f4sub(d0, d2);
            t[1][j] = f4add(d1, d2);
            t[2][j] = f4sub(d2, d1);
            t[3][j] = f4sub(d1, d3);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {                     // V = t B; plane = 4 i + j
            float* o = vdst + (4 * i * TILES + t_id) * KC + 4 * t_q;
            *reinterpret_cast<float4*>(o + 0 * TILES * KC) = f4sub(t[i][0], t[i][2]);
            *reinterpret_cast<float4*>(o + 1 * TILES * KC) = f4add(t[i][1], t[i][2]);
            *reinterpret_cast<float4*>(o + 2 * TILES * KC) = f4sub(t[i][2], t[i][1]);
            *reinterpret_cast<float4*>(o + 3 * TILES * KC) = f4sub(t[i][1], t[i][3]);
        }
    };

    // ---- accumulators: 16 planes x (32 tiles x 32 channels)
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.f;
    const int a_rd = (32 * tg + l31) * KC + 4 * half;     // + plane * TILES * KC
    const int b_rd = (32 * cg + l31) * KC + 4 * half;     // + plane * COB * KC
    auto multiply = [&](const float* vs, const float* us) {
#pragma unroll
        for (int p = 0; p < 16; p++) {
            const float4 av = *reinterpret_cast<const float4*>(vs + p * TILES * KC + a_rd);
            const float4 bv = *reinterpret_cast<const float4*>(us + p * COB * KC + b_rd);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[p], 0, 0, 0);
        }
    };

    // ---- pipeline over channel chunks.  Chunk k lives in v[k & 1], u[k & 1]; its raw patch passed through raw[k & 1].
    // Step k: transform raw[(k+1) & 1] -> v[(k+1) & 1] and multiply chunk k (independent instruction streams the scheduler
    // interleaves), then -- once every wavefront is done with them -- refill u[k & 1] (LDS-DMA) and raw[k & 1] (from the
    // registers loaded one step earlier) with chunk k+2 and start the global loads of chunk k+3.
    const int nk = g.Cin / KC;
    load_raw(0);
    issue_u(0, u0);
    store_raw(raw0);
    if (nk > 1) load_raw(1);
    __syncthreads();
    transform(raw0, v0);
    if (nk > 1) { issue_u(1, u1); store_raw(raw1); }
    if (nk > 2) load_raw(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto step = [&](int kc, const float* vc, const float* uc, const float* raw_next, float* v_next, float* u_refill,
                    float* raw_refill) {
        if (kc + 1 < nk) transform(raw_next, v_next);
        multiply(vc, uc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // u of chunk kc+1 (issued one step ago) and the raw registers of kc+2
        __syncthreads();                                   // every wavefront is done with vc, uc and raw_next
        if (kc + 2 < nk) {
            issue_u(kc + 2, u_refill);
            store_raw(raw_refill);
            if (kc + 3 < nk) load_raw(kc + 3);
        }
        __syncthreads();                                   // v_next and the refilled raw patch are visible
    };
    for (int kc = 0; kc < nk; kc += 2) {
        step(kc, v0, u0, raw1, v1, u0, raw0);
        if (kc + 1 < nk) step(kc + 1, v1, u1, raw0, v0, u1, raw1);
    }

    // ---- output transform + store.  Lane: channel co0 + 32 cg + l31; register r: tile 32 tg + acc_row(r, half)
    const int co = co0 + 32 * cg + l31;
    if (co >= g.Cout) return;
    const float bias = g.bias ? g.bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int tl = 32 * tg + acc_row(r, half);
        const int im = tl / tiles_per_img, rr = tl - im * tiles_per_img;
        const int ty = rr / g.tbw, tx = rr - ty * g.tbw;
        const int n = n0 + im, h = 2 * (th0 + ty), w = 2 * (tw0 + tx);
        if (im >= g.nb || n >= g.N || h >= g.H || w >= g.W) continue;
        float m[16];
#pragma unroll
        for (int p = 0; p < 16; p++) m[p] = acc[p][r];
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {                     // A^T M
            s0[j] = m[0 + j] + m[4 + j] + m[8 + j];
            s1[j] = m[4 + j] - m[8 + j] - m[12 + j];
        }
        float o00 = s0[0] + s0[1] + s0[2] + bias, o01 = s0[1] - s0[2] - s0[3] + bias;
        float o10 = s1[0] + s1[1] + s1[2] + bias, o11 = s1[1] - s1[2] - s1[3] + bias;
        if (g.neg_slope != 1.f) {
            o00 = o00 >= 0.f ? o00 : o00 * g.neg_slope; o01 = o01 >= 0.f ? o01 : o01 * g.neg_slope;
            o10 = o10 >= 0.f ? o10 : o10 * g.neg_slope; o11 = o11 >= 0.f ? o11 : o11 * g.neg_slope;
        }
        float* yp = g.y + (((size_t)n * g.H + h) * g.W + w) * g.Cout + co;
        yp[0] = o00;
        yp[g.Cout] = o01;
        yp[(size_t)g.W * g.Cout] = o10;
        yp[(size_t)g.W * g.Cout + g.Cout] = o11;
    }
}

// U[16][Cout'][Cin'] = G g G^T of every filter.  w: torch conv weight [Co, Ci, 3, 3] in channels_last memory =
// [Co][kh][kw][Ci].  transpose_flip = 0: forward (Cout' = Co, Cin' = Ci, g = w[co, ci]);  1: input gradient
// (Cout' = Ci, Cin' = Co, g[kh][kw] = w[co, ci][2 - kh][2 - kw]).
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int transpose_flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * Ci) return;
    const int o = transpose_flip ? i / Co : i / Ci;       // row of U (output-channel role)
    const int c = transpose_flip ? i % Co : i % Ci;       // column of U (reduction role)
    const int co = transpose_flip ? c : o, ci = transpose_flip ? o : c;
    float gk[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int kh = transpose_flip ? 2 - a : a, kw = transpose_flip ? 2 - b : b;
            gk[a][b] = w[((size_t)(co * 3 + kh) * 3 + kw) * Ci + ci];
        }
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; b++) {                         // G g
        t[0][b] = gk[0][b];
        t[1][b] = 0.5f * (gk[0][b] + gk[1][b] + gk[2][b]);
        t[2][b] = 0.5f * (gk[0][b] - gk[1][b] + gk[2][b]);
        t[3][b] = gk[2][b];
    }
    const int rows = transpose_flip ? Ci : Co, cols = transpose_flip ? Co : Ci;
#pragma unroll
    for (int a = 0; a < 4; a++) {                         // (G g) G^T
        const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
        float* dst = U + ((size_t)(4 * a) * rows + o) * cols + c;
        dst[0] = u0;
        dst[(size_t)rows * cols] = u1;
        dst[2 * (size_t)rows * cols] = u2;
        dst[3 * (size_t)rows * cols] = u3;
    }
}

}  // namespace

extern "C" int scp_wino_filter_transform(const float* w, float* U, int Co, int Ci, int transpose_flip, void* stream) {
    if (Co <= 0 || Ci <= 0) return scp::fail(hipErrorInvalidValue, "wino_filter_transform: empty filter");
    const int n = Co * Ci;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, U, Co, Ci,
                       transpose_flip);
    return scp::check_launch("wino_filter_transform");
}

extern "C" int scp_wino_conv3x3(const float* x, const float* U, const float* bias, float* y, int N, int H, int W, int Cin,
                                int Cout, float negative_slope, void* stream) {
    if (N <= 0) return 0;
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return scp::fail(hipErrorInvalidValue, "wino_conv3x3: H and W must be even");
    if (Cin % KC != 0 || Cin < KC) return scp::fail(hipErrorInvalidValue, "wino_conv3x3: Cin must be a multiple of 8");
    if (Cout % 4 != 0) return scp::fail(hipErrorInvalidValue, "wino_conv3x3: Cout must be a multiple of 4");
    if ((size_t)16 * Cout * Cin >= (1ull << 31)) return scp::fail(hipErrorInvalidValue, "wino_conv3x3: filter too large");
    WinoArgs g{};
    g.x = x; g.U = U; g.bias = bias; g.y = y;
    g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    const int th = H / 2, tw = W / 2;
    g.tbw = tw >= 8 ? 8 : tw;
    g.tbh = th >= 8 ? 8 : th;
    // small images: several per block (nb * tbh * tbw <= 64; powers of two for the shapes of the encoder)
    g.ph = 2 * g.tbh + 2; g.pw = 2 * g.tbw + 2;
    g.nb = TILES / (g.tbh * g.tbw);
    if (g.nb > MAX_PATCH / (g.ph * g.pw)) g.nb = MAX_PATCH / (g.ph * g.pw);
    if (g.nb < 1) g.nb = 1;
    if (g.nb * g.ph * g.pw > MAX_PATCH || g.nb * g.tbh * g.tbw > TILES)
        return scp::fail(hipErrorInvalidValue, "wino_conv3x3: unsupported spatial size");
    g.blocks_h = (th + g.tbh - 1) / g.tbh; g.blocks_w = (tw + g.tbw - 1) / g.tbw;
    g.img_groups = (N + g.nb - 1) / g.nb;
    g.co_blocks = (Cout + COB - 1) / COB;
    g.neg_slope = negative_slope;
    const long blocks = (long)g.img_groups * g.blocks_h * g.blocks_w * g.co_blocks;
    hipLaunchKernelGGL(wino_conv_kernel, dim3((unsigned)blocks), dim3(THREADS), 0, static_cast<hipStream_t>(stream), g);
    return scp::check_launch("wino_conv3x3");
}
