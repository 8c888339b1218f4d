// self-corr-pose_amd/csrc/conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution on NHWC fp32 activations as an implicit GEMM on
// the gfx950 fp32 matrix cores: the stride-1 3x3 layers of the image encoder (ResNet18 BasicBlocks and the U-decoder's
// conv units, model/module/network/image_encoder.py:119-193), forward and -- with the weights transposed and flipped by the
// caller -- input gradient.
//
// GEMM view: C[M = N*H*W pixels][Cout] = A[M][K = 9 Cin] W[Cout][K]^T with K ordered (tap, channel).  In NHWC the 16
// consecutive channels of one tap of one pixel are 64 contiguous bytes -- exactly one row of an LDS-DMA piece of the ViT GEMM
// (csrc/vit_gemm.hip) -- and the channels_last storage of a [Cout, Cin, 3, 3] weight tensor IS [Cout][ky][kx][Cin], i.e. the
// K-contiguous W operand.  So this is the GEMM kernel (128 x 128 x 16 tiles, 4 wavefronts of 2 x 2 MFMA 32x32x2 tiles, 3-stage
// LDS-DMA ring with XOR-swizzled 16-byte slots, persistent XCD-aware tile walk, hand-written ds_read_b128) with one change:
// the A source address of a row is its pixel's address plus a wavefront-uniform tap offset, and rows whose tap falls outside
// the image read a 64-byte block of zeros instead (per-row 9-bit validity mask, computed once per tile).
// Nothing is unfolded (no im2col buffer): the nine taps of a pixel re-read the same input rows through L2.
// Roofline: bound = fp32 MFMA (157.3 TFLOP/s); algorithmic flops 2 M Cout 9 Cin; algorithmic bytes 4 (M Cin + 9 Cin Cout + M Cout).
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int THREADS = 256;
constexpr int TILE_FLOATS = BM * BK;            // one operand tile of one stage (8 KiB)

#define SCP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SCP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct ConvArgs {
    const float* x;        // [N, H, W, Cin]
    const float* w;        // [Cout, 3, 3, Cin]
    const float* bias;     // [Cout] or nullptr
    const float* zeros;    // >= 64 bytes of zeros
    float* y;              // [N, H, W, Cout]
    int H, W, Cin, Cout, M, K;
    int chunks_per_tap;    // Cin / 16
    int nblk_n, panels, per_xcd;
};

__global__ __launch_bounds__(THREADS, 3) void conv3x3_kernel(const ConvArgs g) {
    // one LDS object per (operand, stage): see csrc/vit_gemm.hip
    __shared__ __attribute__((aligned(16))) float a_lds0[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float a_lds1[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float a_lds2[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float w_lds0[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float w_lds1[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float w_lds2[TILE_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int prow = lane >> 2, pslot = lane & 3;

    // tile order: as the GEMM -- tile t runs on XCD t % 8 and the N-blocks of one 128-pixel panel run back to back on one XCD
    const int total = g.per_xcd * 8;
    struct Tile { int m0, n0; bool ok; };
    auto tile_of = [&](int t) {
        Tile x;
        const int lid = (t & 7) * g.per_xcd + (t >> 3);
        x.ok = lid < g.panels * g.nblk_n;
        const int bm = lid / g.nblk_n;
        x.m0 = bm * BM;
        x.n0 = (lid - bm * g.nblk_n) * BN;
        return x;
    };
    // Per wavefront and stage: A pieces 2w, 2w+1 and W pieces alike (16 rows x 64 B each).  Loop-invariant per tile and lane:
    // the pixel's element offset (+ swizzled 16-byte slot), the validity of its nine taps, the weight row's offset.
    unsigned a_pix[2], w_off[2], tap_ok[2];
    const unsigned zero_off = 4u * pslot;
    auto set_offsets = [&](const Tile& t) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = 16 * (2 * wave + i) + prow;                      // tile row 0..127
            const int chunk = pslot ^ ((r >> 2) & 3);
            const int p = t.m0 + r;                                         // pixel (M is a multiple of BM)
            const int img = p / (g.H * g.W), rem = p - img * (g.H * g.W);
            const int yy = rem / g.W, xx = rem - yy * g.W;
            a_pix[i] = (unsigned)p * (unsigned)g.Cin + 4u * chunk;
            unsigned ok = 0;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int y2 = yy + tap / 3 - 1, x2 = xx + tap % 3 - 1;
                if (y2 >= 0 && y2 < g.H && x2 >= 0 && x2 < g.W) ok |= 1u << tap;
            }
            tap_ok[i] = ok;
            w_off[i] = (unsigned)min(t.n0 + r, g.Cout - 1) * (unsigned)g.K + 4u * chunk;
        }
    };
    // K chunk kc = channels [16 c, 16 c + 16) of tap kc / chunks_per_tap; chunks are issued in increasing order within a tile
    int next_tap = 0, next_c = 0;
    auto issue_stage = [&](int kc, float* a_dst, float* w_dst) {
        const int tap = next_tap;
        const int delta = ((tap / 3 - 1) * g.W + (tap % 3 - 1)) * g.Cin + 16 * next_c;      // wavefront-uniform, may be negative
        if (++next_c == g.chunks_per_tap) { next_c = 0; next_tap++; }
        const float* wp = g.w + kc * BK;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const bool ok = (tap_ok[i] >> tap) & 1u;
            const float* src = ok ? g.x + ((long)a_pix[i] + delta) : g.zeros + zero_off;
            __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(src), SCP_LDS_PTR(a_dst + (2 * wave + i) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(wp + w_off[i]), SCP_LDS_PTR(w_dst + (2 * wave + i) * 256), 16, 0, 0);
        }
    };
    const int nk = g.K / BK;

    int t = blockIdx.x;
    Tile cur = tile_of(min(t, total - 1));
    while (t < total && !cur.ok) { t += gridDim.x; if (t < total) cur = tile_of(t); }
    if (t >= total) return;
    set_offsets(cur);
    issue_stage(0, a_lds0, w_lds0);
    issue_stage(1, a_lds1, w_lds1);            // nk >= 9

    const int row_base = 64 * (wave >> 1), col_base = 64 * (wave & 1);
    // lane's read offsets (floats) inside a stage: rows row_base + 32 i + l31 of A, col_base + 32 j + l31 of W; chunk 2 half + c
    int a_rd[2][2], w_rd[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int ra = row_base + 32 * i + l31, rw = col_base + 32 * i + l31;
            a_rd[i][c] = ra * BK + 4 * ((2 * half + c) ^ ((ra >> 2) & 3));
            w_rd[i][c] = rw * BK + 4 * ((2 * half + c) ^ ((rw >> 2) & 3));
        }

    while (true) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        auto compute_stage = [&](const float* as, const float* ws) {
            const unsigned a_base = (unsigned)(size_t)SCP_LDS_PTR(as), w_base = (unsigned)(size_t)SCP_LDS_PTR(ws);
            f32x4 av[2][2], wv[2][2];
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(av[i][c]) : "v"(a_base + 4u * (unsigned)a_rd[i][c]));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(wv[i][c]) : "v"(w_base + 4u * (unsigned)w_rd[i][c]));
                }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(wv[0][0]), "+v"(wv[0][1]),
                           "+v"(wv[1][0]), "+v"(wv[1][1]));
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].x, wv[j][c].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].y, wv[j][c].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].z, wv[j][c].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c].w, wv[j][c].w, acc[i][j], 0, 0, 0);
                    }
        };
        // three-stage ring, prefetch distance two chunks, bare s_barrier (csrc/vit_gemm.hip explains each choice)
        auto step = [&](int kc, const float* as, const float* ws, float* a_next, float* w_next) {
            if (kc + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kc + 2 < nk) issue_stage(kc + 2, a_next, w_next);
            compute_stage(as, ws);
        };
        for (int kc = 0; kc < nk; kc += 3) {
            step(kc, a_lds0, w_lds0, a_lds2, w_lds2);
            if (kc + 1 < nk) step(kc + 1, a_lds1, w_lds1, a_lds0, w_lds0);
            if (kc + 2 < nk) step(kc + 2, a_lds2, w_lds2, a_lds1, w_lds1);
        }

        // next tile's first two chunks before this tile's epilogue
        int tn = t + gridDim.x;
        Tile nxt = cur;
        bool has_next = false;
        while (tn < total) {
            nxt = tile_of(tn);
            if (nxt.ok) { has_next = true; break; }
            tn += gridDim.x;
        }
        __syncthreads();
        if (has_next) {
            set_offsets(nxt);
            next_tap = 0;
            next_c = 0;
            issue_stage(0, a_lds0, w_lds0);
            issue_stage(1, a_lds1, w_lds1);
        }

        // epilogue: lane holds y[pixel m][channel n = n0 + col_base + 32 j + l31] for 16 rows m per 32 x 32 tile
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int mb = cur.m0 + row_base + 32 * i;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int n = cur.n0 + col_base + 32 * j + l31;
                const bool n_ok = n < g.Cout;
                const float b = (g.bias && n_ok) ? g.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = mb + acc_row(r, half);
                    if (n_ok) g.y[(size_t)m * g.Cout + n] = acc[i][j][r] + b;
                }
            }
        }
        if (!has_next) break;
        t = tn;
        cur = nxt;
    }
}

int resident_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus <= 0)
            cus = 256;
        slots = max(8, (3 * cus) & ~7);
    }
    return slots;
}

}  // namespace

extern "C" int scp_conv3x3_nhwc_forward(const float* x, const float* w, const float* bias, const float* zeros64, float* y, int N,
                                        int H, int W, int Cin, int Cout, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return scp::fail(hipErrorInvalidValue, "conv3x3: empty problem");
    if (!x || !w || !y || !zeros64) return scp::fail(hipErrorInvalidValue, "conv3x3: null argument");
    const long M = (long)N * H * W;
    if (Cin % 16 != 0) return scp::fail(hipErrorInvalidValue, "conv3x3: Cin must be a multiple of 16");
    if (M % BM != 0) return scp::fail(hipErrorInvalidValue, "conv3x3: N*H*W must be a multiple of 128");
    if (M * Cin >= (1l << 32) || (long)Cout * 9 * Cin >= (1l << 32) || M * Cout >= (1l << 32))
        return scp::fail(hipErrorInvalidValue, "conv3x3: tensor larger than 2^32 elements");
    ConvArgs g{};
    g.x = x; g.w = w; g.bias = bias; g.zeros = zeros64; g.y = y;
    g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.M = (int)M; g.K = 9 * Cin;
    g.chunks_per_tap = Cin / BK;
    g.nblk_n = (Cout + BN - 1) / BN;
    g.panels = (int)(M / BM);
    g.per_xcd = (g.panels * g.nblk_n + 7) / 8;
    const int total = g.per_xcd * 8;
    hipLaunchKernelGGL(conv3x3_kernel, dim3(min(total, resident_slots())), dim3(THREADS), 0, static_cast<hipStream_t>(stream), g);
    return scp::check_launch("conv3x3");
}
