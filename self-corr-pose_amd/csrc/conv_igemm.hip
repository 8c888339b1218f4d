// self-corr-pose_amd/csrc/conv_igemm.hip -- the image encoder's 3x3 / 1x1 convolutions on NHWC fp32 activations as implicit
// GEMMs on the gfx950 fp32 matrix cores: forward, input gradient (the same kernel on transformed weights) -- the weight
// gradient is csrc/conv_wgrad.hip.
//
// Replaces MIOpen for model/module/network/image_encoder.py:119-193 (ResNet18 BasicBlocks of the trunk, the U-decoder's
// conv units of net_blocks.py:336-359), called twice per step (encoder.py:29-37, correspondence.py:91).
//
// GEMM view: y[M = N Ho Wo pixels][Cout] = A[M][K = taps x Cin] W[Cout][K]^T with K ordered (tap, channel).  In NHWC the 16
// consecutive channels of one tap of one pixel are 64 contiguous bytes -- one LDS-DMA row of csrc/gemm_core.h -- and the
// channels_last storage of a [Cout, Cin, kh, kw] weight tensor IS [Cout][ky][kx][Cin], the K-contiguous W operand.  So this is
// the GEMM core with an A source that gathers: the source of a row is its pixel's address plus a wavefront-uniform tap offset.
// Taps that fall outside the image are not branched on: the A tile is loaded through a BUFFER descriptor
// (buffer_load_dwordx4 ... lds) and their lanes get an offset beyond the descriptor's size, for which the hardware returns
// zeros -- no zero page, no im2col, nothing unfolded; the nine taps of a pixel re-read the same input rows through L2.
// Stride 2 (the first block of layer2..4) only changes the pixel -> address map.
//
// Epilogues: raw output (BatchNorm follows); + bias + LeakyReLU (decoder units, net_blocks.py:336-359 with_bn=False);
// optionally the per-channel sum and sum of squares of the tile's rows, so that the BatchNorm that follows does not have to
// read the activation again for its statistics (csrc/batchnorm.hip folds the per-tile partials in a fixed order).
// Tile shapes (all 4 wavefronts, 2 workgroups per CU): 256 x 64 for the 64-channel layers (131072 pixels -> 512 tiles),
// 64 x 128 and 64 x 64 for the deeper, narrower-in-pixels layers so that every layer launches >= 256 tiles.
// Roofline: bound = fp32 MFMA (157.3 TFLOP/s nominal; the clock the part sustains under this load is ~1.95 GHz = 128 TFLOP/s,
// tools/probes/gemm_v3.hip); algorithmic flops 2 M Cout taps Cin; algorithmic bytes 4 (M_in Cin + taps Cin Cout + M Cout).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <utility>

#include "bn_common.h"
#include "gemm_core.h"
#include "gemm_core_split.h"
#include "scp_common.h"
#include "scp_hip.h"

namespace {

using scp::f32x16;

struct ConvArgs {
    const float* x;        // [N, H, W, Cin]
    const float* w;        // [Cout, taps, Cin]
    const void* w_split;   // or its planes [3][Cout][taps * Cin] bf16 (scp_split_bf16x3): main loop of csrc/gemm_core_split.h
    const float* bias;     // [Cout] or nullptr
    float* y;              // [N, Ho, Wo, Cout]
    float* partials;       // [2][tiles_m][Cout] (sums, then sums of squares) or nullptr
    float* ypart;          // split-K: [ksplit][M][Cout] raw partial outputs (folded by conv_splitk_fold_kernel)
    int ksplit, nk_split;  // split-K: workgroups per tile, chunks per workgroup
    unsigned* ticket;      // with partials: the last workgroup folds them and finalises the BatchNorm statistics (fin)
    scp_bn::FwdFinalize fin;
    int H, W, Ho, Wo, Cin, Cout, M, K;
    int stride, lg_cpt;    // chunks per tap = Cin / 16 = 1 << lg_cpt
    int nblk_n, tiles_m;   // tiles_m: row tiles of the statistics partials (= the kernel's, or the fold kernel's with split-K)
    int tiles_m_kernel;    // row tiles of the convolution kernel's grid
    unsigned x_bytes;
    float slope;
    int scatter2;          // 1x1 input gradient of a stride-2 projection: row m = dy pixel (a, b) -> dx pixel (2a, 2b), its three neighbours zeros
};

// one LDS-DMA piece through a buffer descriptor: lanes whose offset is beyond the descriptor read zeros
__device__ __forceinline__ void bufload16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}

// A-tile source of the implicit GEMM (shared by the fp32 and the split main loop): tile row r = output pixel, chunk kc = 16
// channels of one tap
template <class CFG, int TAPS>
struct ConvASource {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned a_pix[CFG::A_PER], tap_ok[CFG::A_PER];
    int Win, Cin, lg_cpt;

    __device__ __forceinline__ void set(const ConvArgs& g, int m0, int wave, int lane) {
        const int prow = lane >> 2, pslot = lane & 3;
        const int chunk = pslot ^ ((prow >> 2) & 3);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)g.x_bytes, 0x00020000);
        Win = g.W; Cin = g.Cin; lg_cpt = g.lg_cpt;
        const int hw = g.Ho * g.Wo;
#pragma unroll
        for (int i = 0; i < CFG::A_PER; i++) {
            const int r = 16 * (wave * CFG::A_PER + i) + prow;
            const int p = min(m0 + r, g.M - 1);
            const int img = p / hw, rem = p - img * hw;
            const int yo = rem / g.Wo, xo = rem - yo * g.Wo;
            const int yi = yo * g.stride, xi = xo * g.stride;          // centre tap
            a_pix[i] = ((unsigned)((img * g.H + yi) * g.W + xi) * (unsigned)g.Cin + 4u * chunk) * 4u;
            unsigned ok = 0;
            if (TAPS == 9) {
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const int y2 = yi + tap / 3 - 1, x2 = xi + tap % 3 - 1;
                    if (y2 >= 0 && y2 < g.H && x2 >= 0 && x2 < g.W) ok |= 1u << tap;
                }
            } else {
                ok = 1u;
            }
            tap_ok[i] = ok;
        }
    }
    template <int I>
    __device__ __forceinline__ void issue(int kc, unsigned stage_lds, int wave) const {
#ifdef SCP_PROBE_NO_DMA            // tools/probes: the main loop without its global -> LDS traffic
        if (kc >= 2) return;
#endif
        const int tap = kc >> lg_cpt, c = kc - (tap << lg_cpt);
        int delta = 64 * c;
        if (TAPS == 9) {
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;                   // tap / 3, tap % 3 for tap < 9
            delta += ((ky - 1) * Win + (kx - 1)) * Cin * 4;
        }
        const bool ok = (tap_ok[I] >> tap) & 1u;
        const unsigned voff = ok ? a_pix[I] + (unsigned)delta : 0x80000000u;      // beyond the descriptor: zeros
        bufload16(voff, rsrc, stage_lds + (unsigned)(wave * CFG::A_PER + I) * 1024u);
    }
};

// both operands of the fp32 main loop (csrc/gemm_core.h)
template <class CFG, int TAPS>
struct ConvSource {
    ConvASource<CFG, TAPS> a;
    const char* w_base;
    unsigned w_off[CFG::W_PER];

    __device__ __forceinline__ void set(const ConvArgs& g, int m0, int n0, int wave, int lane) {
        const int prow = lane >> 2, pslot = lane & 3;
        const int chunk = pslot ^ ((prow >> 2) & 3);
        a.set(g, m0, wave, lane);
        w_base = reinterpret_cast<const char*>(g.w);
#pragma unroll
        for (int i = 0; i < CFG::W_PER; i++) {
            const int r = 16 * (wave * CFG::W_PER + i) + prow;
            w_off[i] = ((unsigned)min(n0 + r, g.Cout - 1) * (unsigned)g.K + 4u * chunk) * 4u;
        }
    }
    template <int I>
    __device__ __forceinline__ void issue(int kc, unsigned stage_lds, int wave) const {
        if constexpr (I < CFG::A_PER) {
            a.template issue<I>(kc, stage_lds, wave);
        } else {
            scp::glds16(w_off[I - CFG::A_PER], w_base + (size_t)kc * (CFG::BK * 4),
                        stage_lds + CFG::W_BASE_BYTES + (unsigned)(wave * CFG::W_PER + (I - CFG::A_PER)) * 1024u);
        }
    }
};

enum { EPI_RAW = 0, EPI_BIAS_LEAKY = 1 };

// last workgroup of a launch: per-tile column sums -> batch statistics of the BatchNorm that follows (fp64 fold in tile order)
__device__ __forceinline__ void finalize_statistics(const ConvArgs& g, float* lds) {
    if (threadIdx.x == 0 && g.fin.batches_tracked) *g.fin.batches_tracked += 1;
    const int tc_n = g.Cout / 4;
    double sa[4], sb[4];
    __syncthreads();
    if (!scp_bn::fold_partials(g.partials, g.partials + (size_t)g.tiles_m * g.Cout, g.tiles_m, g.Cout, tc_n, sa, sb,
                               reinterpret_cast<float4*>(lds)))
        return;
    const int tc = threadIdx.x % tc_n;
#pragma unroll
    for (int i = 0; i < 4; i++) scp_bn::finalize_channel(g.fin, 4 * tc + i, true, sa[i], sb[i]);
}

// the split twin of a tile shape: same wavefront grid and MFMA tiles per wavefront
// (round 4: its W planes are TILED -- csrc/gemm_core_split.h -- every LDS-DMA piece of the weight one contiguous KiB instead of
// 32 rows x 32 B = 32 cache lines: the weight pieces were 3/4 of the main loop's address traffic)
template <class CFG>
using SplitOf = scp::SplitCfg<CFG::WM, CFG::WN, CFG::NWM, CFG::NWN, CFG::MINBLK, 3, (CFG::WM * CFG::WN <= 4), false, true>;

template <class FCFG, int TAPS, int EPI, bool STATS, bool SPLIT>
__global__ __launch_bounds__(FCFG::THREADS, 2) void conv_igemm_kernel(const ConvArgs g) {
    using CFG = std::conditional_t<SPLIT, SplitOf<FCFG>, FCFG>;
    if constexpr (SPLIT) scp::claim_vgprs_for_tile<FCFG::WM, FCFG::WN>();       // bf16 MFMAs: the wavefronts tile the SIMD's register file (scp_common.h)
    __shared__ __attribute__((aligned(16))) float lds[CFG::LDS_BYTES / 4];
    using Core = std::conditional_t<SPLIT, scp::SplitGemmCore<SplitOf<FCFG>, ConvASource<SplitOf<FCFG>, TAPS>>,
                                    scp::GemmCore<FCFG, ConvSource<FCFG, TAPS>>>;
    // tile order: workgroup t runs on XCD t % 8; consecutive tiles of one XCD walk the column blocks of one pixel panel
    // split-K (split main loop, deep layers): ksplit consecutive workgroups share a tile, each its own range of K chunks; they
    // write raw partial tiles, conv_splitk_fold_kernel adds them and applies the epilogue / statistics
    const int ksplit = SPLIT ? g.ksplit : 1;
    const int total = g.tiles_m_kernel * g.nblk_n * ksplit;
    const int per_xcd = (total + 7) >> 3;
    const int lid0 = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lid0 >= total) {
        // a padding workgroup still takes its ticket (the last arrival is counted over the whole grid); it can be the last
        if (STATS && g.ticket && scp_bn::last_block_arrived(g.ticket, reinterpret_cast<int*>(lds))) finalize_statistics(g, lds);
        return;
    }
    const int lid = lid0 / ksplit, ks = lid0 - lid * ksplit;
    const int bm = lid / g.nblk_n, bn = lid - bm * g.nblk_n;
    const int m0 = bm * CFG::BM, n0 = bn * CFG::BN;
    Core core(lds);
    float* yout = g.y;
    if constexpr (SPLIT) {
        core.asrc.set(g, m0, core.wave, core.lane);
        core.set_w_rows(g.w_split, g.Cout, g.K, [&](int r) { return min(n0 + r, g.Cout - 1); });
        if (ksplit > 1) {
            core.kc0 = ks * g.nk_split;
            yout = g.ypart + (size_t)ks * g.M * g.Cout;
        }
    } else {
        core.src.set(g, m0, n0, core.wave, core.lane);
    }
    typename Core::Acc acc;
    core.run(acc, (SPLIT && ksplit > 1) ? g.nk_split : g.K / CFG::BK);

    const int half = core.lane >> 5, l31 = core.lane & 31;
    float csum[CFG::WN], csq[CFG::WN];
#pragma unroll
    for (int j = 0; j < CFG::WN; j++) csum[j] = csq[j] = 0.f;
#pragma unroll
    for (int i = 0; i < CFG::WM; i++) {
        const int mb = m0 + core.row_base() + 32 * i;
#pragma unroll
        for (int j = 0; j < CFG::WN; j++) {
            const int n = n0 + core.col_base() + 32 * j + l31;
            const bool n_ok = n < g.Cout;
            float b = 0.f;
            if (EPI == EPI_BIAS_LEAKY) b = g.bias[min(n, g.Cout - 1)];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + scp::acc_row(r, half);
                float v = acc.t[i * CFG::WN + j][r];
                const bool live = m < g.M && n_ok;
                if (STATS && live) {
                    csum[j] += v;
                    csq[j] += v * v;
                }
                if (EPI == EPI_BIAS_LEAKY) {
                    v += b;
                    v = v > 0.f ? v : v * g.slope;
                }
                if (EPI == EPI_RAW && !STATS && TAPS == 1 && g.scatter2) {
                    // dx of a 1x1 / stride-2 convolution: the product lands on the even pixel, the other three of its 2 x 2 cell
                    // get no contribution -- written here as zeros (no memset of dx, no strided copy)
                    if (live) {
                        const int hw = g.Ho * g.Wo, img = m / hw, rem = m - img * hw, a = rem / g.Wo, b = rem - a * g.Wo;
                        float* o = yout + (((size_t)img * 2 * g.Ho + 2 * a) * (2 * g.Wo) + 2 * b) * g.Cout + n;
                        o[0] = v;
                        o[g.Cout] = 0.f;
                        o[(size_t)2 * g.Wo * g.Cout] = 0.f;
                        o[(size_t)(2 * g.Wo + 1) * g.Cout] = 0.f;
                    }
                } else if (live) {
                    yout[(size_t)m * g.Cout + n] = v;
                }
            }
        }
    }
    if (STATS) {
        // column sums of this tile's rows: the two half-waves hold different rows of the same columns; the NWM wavefronts
        // stacked along M are folded through LDS (the ring is idle now) in a fixed order
        __syncthreads();
        float* red = lds;                                   // [NW][WN][2][32]
#pragma unroll
        for (int j = 0; j < CFG::WN; j++) {
            const float s = csum[j] + __shfl_xor(csum[j], 32), q = csq[j] + __shfl_xor(csq[j], 32);
            if (half == 0) {
                red[((core.wave * CFG::WN + j) * 2 + 0) * 32 + l31] = s;
                red[((core.wave * CFG::WN + j) * 2 + 1) * 32 + l31] = q;
            }
        }
        __syncthreads();
        // thread t < BN * 2: (which, column)
        const int t = threadIdx.x;
        if (t < 2 * CFG::BN) {
            const int which = t / CFG::BN, col = t - which * CFG::BN;
            const int wn = col / (32 * CFG::WN), j = (col >> 5) % CFG::WN, l = col & 31;
            float s = 0.f;
#pragma unroll
            for (int wm = 0; wm < CFG::NWM; wm++) s += red[(((wm * CFG::NWN + wn) * CFG::WN + j) * 2 + which) * 32 + l];
            const int n = n0 + col;
            // write-through (visible to the folding workgroup on another XCD once complete; csrc/bn_common.h)
            if (n < g.Cout)
                __hip_atomic_store(g.partials + ((size_t)which * g.tiles_m + bm) * g.Cout + n, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (g.ticket && scp_bn::last_block_arrived(g.ticket, reinterpret_cast<int*>(lds))) finalize_statistics(g, lds);
    }
}

// ---- input gradient of the 3x3 / stride-2 / pad-1 convolutions (first block of layer2..4) -----------------------------------------
// dx[img][yi][xi][ci] = sum over (ky, kx, co) with yi + 1 - ky and xi + 1 - kx EVEN of dy[img][(yi+1-ky)/2][(xi+1-kx)/2][co] w[co][ci][ky][kx].
// Which taps contribute depends only on the PARITY of (yi, xi): even -> the centre tap alone (k = 1), odd -> k = 0 and k = 2.  So the
// input pixels fall into four classes (py, px) with 1, 2, 2 and 4 taps; each class is an implicit GEMM over the dy grid
//     rows = (img, a, b) of the dy grid  <->  input pixel (2a + py, 2b + px),    K = (taps of the class) x Cout,
// a tap reading dy at (a + [k == 0], b + [k == 0]) -- beyond the grid: zeros through the buffer descriptor, as in the forward kernel.
// 9 tap-products per 4 pixels instead of the 36 a "forward kernel on a zero-stuffed dy" would do, and no zeros written or read.
// The W operand is the `dgrad` plane set of conv_weight_planes_kernel (rows = ci, K = (flipped tap, co)): the class walks the chunks
// of ITS taps (SplitGemmCore's w_chunk hook).  All four classes in one launch (see the kernel for how they are dealt to workgroups).
template <class CFG>
struct Dgrad2ASource {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned a_pix[CFG::A_PER], tap_ok[CFG::A_PER];
    int Wo, C, lg_cpt, py, px;

    // tap t of the class -> (ky, kx): py ? {0, 2}[t / nx] : 1
    __device__ __forceinline__ void tap_kykx(int tap, int& ky, int& kx) const {
        const int iy = px ? tap >> 1 : tap, ix = px ? tap & 1 : 0;
        ky = py ? 2 * iy : 1;
        kx = px ? 2 * ix : 1;
    }
    __device__ __forceinline__ void set(const ConvArgs& g, int py_, int px_, int m0, int wave, int lane) {
        const int prow = lane >> 2, pslot = lane & 3;
        const int chunk = pslot ^ ((prow >> 2) & 3);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)g.x_bytes, 0x00020000);
        Wo = g.W; C = g.Cin; lg_cpt = g.lg_cpt; py = py_; px = px_;
        const int hw = g.H * g.W, ntaps = (1 + py) * (1 + px);
#pragma unroll
        for (int i = 0; i < CFG::A_PER; i++) {
            const int r = 16 * (wave * CFG::A_PER + i) + prow;
            const int p = min(m0 + r, g.M - 1);
            const int img = p / hw, rem = p - img * hw;
            const int a = rem / g.W, b = rem - a * g.W;
            a_pix[i] = ((unsigned)p * (unsigned)g.Cin + 4u * chunk) * 4u;
            unsigned ok = 0;
            for (int tap = 0; tap < ntaps; tap++) {
                int ky, kx;
                tap_kykx(tap, ky, kx);
                if (a + (ky == 0) < g.H && b + (kx == 0) < g.W) ok |= 1u << tap;
            }
            tap_ok[i] = ok;
        }
    }
    template <int I>
    __device__ __forceinline__ void issue(int kc, unsigned stage_lds, int wave) const {
        const int tap = kc >> lg_cpt, c = kc - (tap << lg_cpt);
        int ky, kx;
        tap_kykx(tap, ky, kx);
        const int delta = 64 * c + ((ky == 0) * Wo + (kx == 0)) * C * 4;
        const bool ok = (tap_ok[I] >> tap) & 1u;
        const unsigned voff = ok ? a_pix[I] + (unsigned)delta : 0x80000000u;
        bufload16(voff, rsrc, stage_lds + (unsigned)(wave * CFG::A_PER + I) * 1024u);
    }
    __device__ __forceinline__ int w_chunk(int kc) const {
        const int tap = kc >> lg_cpt, c = kc - (tap << lg_cpt);
        int ky, kx;
        tap_kykx(tap, ky, kx);
        return (((2 - ky) * 3 + (2 - kx)) << lg_cpt) + c;
    }
};

// ConvArgs here: x = dy [N, H = Ho, W = Wo, Cin = conv Cout], y = dx [N, 2 Ho, 2 Wo, Cout = conv Cin], M = N Ho Wo (rows per class)
template <class FCFG>
__global__ __launch_bounds__(FCFG::THREADS, 2) void conv_dgrad_s2_kernel(const ConvArgs g) {
    using CFG = SplitOf<FCFG>;
    if constexpr (FCFG::WM * FCFG::WN >= 4) scp::claim_vgprs<256>();            // 180 - 196 registers: the 2 x 256 class (scp_common.h)
    else scp::claim_vgprs_for_tile<FCFG::WM, FCFG::WN>();
    __shared__ __attribute__((aligned(16))) float lds[CFG::LDS_BYTES / 4];
    using Core = scp::SplitGemmCore<CFG, Dgrad2ASource<CFG>>;
    // two kinds of workgroups per tile of the dy grid: one takes the 4-tap class, the other the 2 + 2 + 1-tap classes one after the
    // other -- 4 and 5 tap-products each, so that a launch of ~one workgroup per CU is balanced (one class per workgroup left the
    // CUs that drew 4-tap tiles with 4x the work of those that drew the centre tap: 64 us against the forward's 46 for equal flops)
    // g.ksplit = workgroups per tile: 2 (the two kinds above) or 4 (one class each: more, smaller workgroups for the layers whose
    // launch would otherwise be one 4-wavefront workgroup per CU -- too few wavefronts to cover the DMA latency)
    const int per_tile = g.ksplit;
    const int total = per_tile * g.tiles_m_kernel * g.nblk_n;
    const int per_xcd = (total + 7) >> 3;
    const int lid0 = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lid0 >= total) return;
    const int sub = __builtin_amdgcn_readfirstlane(lid0 % per_tile), lid = lid0 / per_tile;
    const int ncls = per_tile == 4 ? 1 : (sub ? 3 : 1), cls0 = per_tile == 4 ? 3 - sub : (sub ? 2 : 3);
    const int bm = lid / g.nblk_n, bn = lid - bm * g.nblk_n;
    const int m0 = bm * CFG::BM, n0 = bn * CFG::BN;
    Core core(lds);
    core.set_w_rows(g.w_split, g.Cout, 9 * g.Cin, [&](int r) { return min(n0 + r, g.Cout - 1); });
    const int half = core.lane >> 5, l31 = core.lane & 31;
    const int hw = g.H * g.W;
#pragma nounroll
    for (int c = 0; c < ncls; c++) {
        const int cls = cls0 - c;
        const int py = cls >> 1, px = cls & 1;
        if (c) __syncthreads();                         // the previous class's last fragment reads are done in every wavefront
        core.asrc.set(g, py, px, m0, core.wave, core.lane);
        typename Core::Acc acc;
        core.run(acc, ((1 + py) * (1 + px)) << g.lg_cpt);
        int m0v = m0;
        asm volatile("" : "+v"(m0v));                   // the row -> pixel divisions below are class-invariant: not hoisted (64 x 4 VGPRs)
#pragma unroll
        for (int i = 0; i < CFG::WM; i++) {
            const int mb = m0v + core.row_base() + 32 * i;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + scp::acc_row(r, half);
                const int img = m / hw, rem = m - img * hw;
                const int a = rem / g.W, b = rem - a * g.W;
                const size_t orow = ((size_t)(img * 2 * g.H + 2 * a + py) * (2 * g.W) + 2 * b + px) * g.Cout;
#pragma unroll
                for (int j = 0; j < CFG::WN; j++) {
                    const int n = n0 + core.col_base() + 32 * j + l31;
                    if (m < g.M && n < g.Cout) g.y[orow + n] = acc.t[i * CFG::WN + j][r];
                }
            }
        }
    }
}

// weight [Cout, Cin, k, k] (any strides) -> the split operand planes of both directions in one pass:
//   fwd   rows = Cout, K = (ky, kx, ci)              the W operand of the forward implicit GEMM,
//   dgrad rows = Cin,  K = (ky, kx, co), taps flipped  the W operand of the input gradient (NULL: not wanted)
// both in the TILED plane layout of csrc/gemm_core_split.h ([rows / 32][K / 16][3][32][16] bf16, rows padded to 32)
__global__ void conv_weight_planes_kernel(const float* __restrict__ w, long s_co, long s_ci, long s_ky, long s_kx, int Cout, int Cin,
                                          int k, __bf16* __restrict__ fwd, __bf16* __restrict__ dgrad) {
    const long n = (long)Cout * Cin * k * k;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // fwd order: ((co k + ky) k + kx) Cin + ci
    if (i >= n) return;
    const int ci = (int)(i % Cin);
    long r = i / Cin;
    const int kx = (int)(r % k);
    r /= k;
    const int ky = (int)(r % k), co = (int)(r / k);
    const float v = w[co * s_co + ci * s_ci + ky * s_ky + kx * s_kx];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    const size_t o = scp::tiled_plane_offset(co, (ky * k + kx) * Cin + ci, 0, (k * k * Cin) >> 4);
    fwd[o] = h; fwd[o + 512] = m; fwd[o + 1024] = l;
    if (dgrad) {
        const size_t j = scp::tiled_plane_offset(ci, ((k - 1 - ky) * k + (k - 1 - kx)) * Cout + co, 0, (k * k * Cout) >> 4);
        dgrad[j] = h; dgrad[j + 512] = m; dgrad[j + 1024] = l;
    }
}

// every stale layer of the encoder in ONE launch (Trainer.step refreshes ~27 plane sets right after the optimizer, on the critical
// path before the forward).  A workgroup owns a 32 (co) x 32 (ci) tile of one layer with all its taps: the tile is read in the weight's
// own memory order (contiguous [Cout,Cin,k,k]: 32 k k floats per output channel; channels_last: 32 floats per tap), parked in LDS and
// written out twice -- forward planes with ci running along a lane row, input-gradient planes with co running along it -- so that every
// store instruction of a wavefront fills whole 32-byte rows of a [32 rows][16 k] plane tile (the element-per-thread version scattered the
// input-gradient planes as 2-byte stores 1 KB apart: 160-300 us per step for 20 us worth of traffic).
constexpr int PT = SCP_CONV_PLANES_TILE, PTAPS = 9;
// [tap][co][ci], rows padded by one float and taps by three: conflict-free in both write-out orders, <= 2-way in the read-in order
constexpr int PROW = PT + 1, PTAP = PT * PROW + 3;

// x0, x1 -> their (h, m, l) split, each pair packed into one 32-bit word (element 0 in the low half)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
    const float r0 = x0 - (float)h0, r1 = x1 - (float)h1;
    const __bf16 m0 = (__bf16)r0, m1 = (__bf16)r1;
    const __bf16 l0 = (__bf16)(r0 - (float)m0), l1 = (__bf16)(r1 - (float)m1);
    auto pack = [](__bf16 a, __bf16 b) {
        return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    };
    h = pack(h0, h1); m = pack(m0, m1); l = pack(l0, l1);
}

// one 32 x 32 (co, ci) tile of one layer.  K = the kernel size when it is known at compile time (1, 3: the index arithmetic of the
// read-in loop is then shifts and multiplications), 0 = any size, PTAPS taps per trip through LDS.
// Write-out: a lane owns two neighbouring K positions of a plane row (one 4-byte store per plane), 16 lanes one 32-element... row half:
// every store instruction of a wavefront writes four whole 64-byte runs.  Needs Cin (forward planes) / Cout (input-gradient planes) even,
// which the planes' own K % 16 == 0 rule implies for odd kernel sizes.
template <int K>
__device__ __forceinline__ void planes_tile(const scp_conv_planes_desc& d, float* park, int tile) {
    const int Cout = d.Cout, Cin = d.Cin, k = K ? K : d.ksize, kk = k * k;
    const float* w = reinterpret_cast<const float*>(d.w);
    __bf16* fwd = reinterpret_cast<__bf16*>(d.planes_fwd);
    __bf16* dgrad = reinterpret_cast<__bf16*>(d.planes_dgrad);
    const int tiles_ci = (Cin + PT - 1) / PT;
    const int co0 = (tile / tiles_ci) * PT, ci0 = (tile % tiles_ci) * PT;
    const bool ci_fastest = d.s_ci == 1 && kk > 1;       // channels_last storage: ci runs fastest in memory, then the taps
    const int pair = threadIdx.x & 15, row0 = threadIdx.x >> 4;        // 256 threads = 16 rows of 16 column pairs
    for (int tap0 = 0; tap0 < kk; tap0 += PTAPS) {
        const int nt = K ? K * K : min(PTAPS, kk - tap0);
        for (int idx = threadIdx.x; idx < nt * PT * PT; idx += 256) {
            int co_l, ci_l, t;
            if (ci_fastest) { ci_l = idx % PT; t = (idx / PT) % nt; co_l = idx / (PT * nt); }
            else { t = idx % nt; ci_l = (idx / nt) % PT; co_l = idx / (PT * nt); }
            const int tap = tap0 + t, co = co0 + co_l, ci = ci0 + ci_l;
            park[t * PTAP + co_l * PROW + ci_l] = (co < Cout && ci < Cin) ? w[co * d.s_co + ci * d.s_ci + (tap / k) * d.s_ky + (tap % k) * d.s_kx] : 0.f;
        }
        __syncthreads();
        auto write_tap = [&](int t) {
            const int tap = tap0 + t, ky = tap / k, kx = tap % k;
#pragma unroll
            for (int r = row0; r < PT; r += 16) {
                {   // forward planes: row co, K index (tap, ci)
                    const int co = co0 + r, ci = ci0 + 2 * pair;
                    if (co < Cout && ci < Cin) {
                        unsigned h, m, l;
                        split_pair(park[t * PTAP + r * PROW + 2 * pair], park[t * PTAP + r * PROW + 2 * pair + 1], h, m, l);
                        unsigned* o = reinterpret_cast<unsigned*>(fwd + scp::tiled_plane_offset(co, tap * Cin + ci, 0, (kk * Cin) >> 4));
                        o[0] = h; o[256] = m; o[512] = l;
                    }
                }
                if (dgrad) {   // input-gradient planes: row ci, K index (flipped tap, co)
                    const int ci = ci0 + r, co = co0 + 2 * pair;
                    if (co < Cout && ci < Cin) {
                        unsigned h, m, l;
                        split_pair(park[t * PTAP + (2 * pair) * PROW + r], park[t * PTAP + (2 * pair + 1) * PROW + r], h, m, l);
                        unsigned* o = reinterpret_cast<unsigned*>(
                            dgrad + scp::tiled_plane_offset(ci, ((k - 1 - ky) * k + (k - 1 - kx)) * Cout + co, 0, (kk * Cout) >> 4));
                        o[0] = h; o[256] = m; o[512] = l;
                    }
                }
            }
        };
        if constexpr (K > 0) {
#pragma unroll
            for (int t = 0; t < K * K; t++) write_tap(t);
        } else {
            for (int t = 0; t < nt; t++) write_tap(t);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void conv_weight_planes_batch_kernel(const scp_conv_planes_desc* __restrict__ descs, int n) {
    __shared__ float park[PTAPS * PTAP];
    // the layer of this workgroup: binary search over the ascending block0
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((long long)blockIdx.x >= descs[mid].block0) lo = mid; else hi = mid - 1;
    }
    const scp_conv_planes_desc d = descs[lo];
    const int tile = (int)((long long)blockIdx.x - d.block0);
    if (d.ksize == 3) planes_tile<3>(d, park, tile);
    else if (d.ksize == 1) planes_tile<1>(d, park, tile);
    else planes_tile<0>(d, park, tile);
}

using Cfg256x64 = scp::GemmCfg<2, 2, 4, 1, 2, 2>;     // 64-channel layers at 64 x 64 resolution
using Cfg64x128 = scp::GemmCfg<1, 2, 2, 2, 2, 2>;
using Cfg64x64 = scp::GemmCfg<1, 1, 2, 2, 2, 2>;
using Cfg128x128 = scp::GemmCfg<2, 2, 2, 2, 2, 2>;    // split main loop only: its per-chunk overheads want >= 24 MFMAs per wavefront

template <class CFG, int TAPS>
void launch_cfg(ConvArgs& g, bool leaky, bool stats, hipStream_t st) {
    g.nblk_n = (g.Cout + CFG::BN - 1) / CFG::BN;
    g.tiles_m = g.tiles_m_kernel = (g.M + CFG::BM - 1) / CFG::BM;
    const int total = g.tiles_m * g.nblk_n * (g.w_split && g.ksplit > 1 ? g.ksplit : 1);
    const dim3 grid(((total + 7) >> 3) << 3), block(CFG::THREADS);
    if (g.w_split) {
        if (leaky) {
            if (stats) hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_BIAS_LEAKY, true, true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_BIAS_LEAKY, false, true>), grid, block, 0, st, g);
        } else {
            if (stats) hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_RAW, true, true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_RAW, false, true>), grid, block, 0, st, g);
        }
        return;
    }
    if (leaky) {
        if (stats) hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_BIAS_LEAKY, true, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_BIAS_LEAKY, false, false>), grid, block, 0, st, g);
    } else {
        if (stats) hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_RAW, true, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((conv_igemm_kernel<CFG, TAPS, EPI_RAW, false, false>), grid, block, 0, st, g);
    }
}

// y[m][n] = epilogue(sum over the K splits of ypart[s][m][n]); with STATS also the column sums / sums of squares of the RAW output
// per tile of FOLD_ROWS rows (the partials protocol of the convolution kernel's own epilogue), the last workgroup finalises the batch
// statistics.  grid = row tiles x ceil(Cout / 256); thread = 4 columns x one of 4 row lanes.  (32 rows per workgroup: the layers
// that split K have 2 048 - 8 192 pixels, and with 128-row tiles the launch was 32 - 128 workgroups of a latency-bound loop: 27 us
// each, 0.67 ms per step.)
constexpr int FOLD_ROWS = 32;
template <int EPI, bool STATS>
__global__ __launch_bounds__(256) void conv_splitk_fold_kernel(const ConvArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const int colblocks = (g.Cout + 255) / 256;
    const int bm = blockIdx.x / colblocks, cb = blockIdx.x - bm * colblocks;
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = cb * 256 + 4 * cq;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq2[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < g.Cout) {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_BIAS_LEAKY) {
            const float4 t = *reinterpret_cast<const float4*>(g.bias + n);
            b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
        }
        const size_t split_stride = (size_t)g.M * g.Cout;
        for (int r = rl; r < FOLD_ROWS; r += 4) {
            const int m = bm * FOLD_ROWS + r;
            if (m >= g.M) break;
            const float* src = g.ypart + (size_t)m * g.Cout + n;
            float4 a = *reinterpret_cast<const float4*>(src);
            for (int sidx = 1; sidx < g.ksplit; sidx++) {
                const float4 t = *reinterpret_cast<const float4*>(src + sidx * split_stride);
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (STATS) {
                    cs[i] += v[i];
                    cq2[i] += v[i] * v[i];
                }
                if (EPI == EPI_BIAS_LEAKY) {
                    v[i] += b[i];
                    v[i] = v[i] > 0.f ? v[i] : v[i] * g.slope;
                }
            }
            *reinterpret_cast<float4*>(g.y + (size_t)m * g.Cout + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (STATS) {
        float* red = lds;                                   // [4 row lanes][64 column quads][8]
#pragma unroll
        for (int i = 0; i < 4; i++) {
            red[(rl * 64 + cq) * 8 + i] = cs[i];
            red[(rl * 64 + cq) * 8 + 4 + i] = cq2[i];
        }
        __syncthreads();
        if (rl == 0 && n < g.Cout) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    a += red[(l * 64 + cq) * 8 + i];
                    q += red[(l * 64 + cq) * 8 + 4 + i];
                }
                __hip_atomic_store(g.partials + ((size_t)0 * g.tiles_m + bm) * g.Cout + n + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g.partials + ((size_t)1 * g.tiles_m + bm) * g.Cout + n + i, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (g.ticket && scp_bn::last_block_arrived(g.ticket, reinterpret_cast<int*>(lds))) finalize_statistics(g, lds);
    }
}

// How a layer is run: tile shape (0: 256 x 64, 1: 64 x 128, 2: 64 x 64, 3: 128 x 128) and K splits.
struct ConvPlan { int cfg, ksplit; };
ConvPlan plan_conv(long M, int Cout, int K, bool split) {
    if (Cout <= 64) return {0, 1};                                                 // 256 x 64
    const long t128 = ((M + 63) / 64) * ((Cout + 127) / 128);
    if (split) {
        // the split main loop does 2.7x less matrix-pipe work per chunk while its barriers, DMA waits and operand splits stay:
        // 128 x 128 (24 MFMAs per wavefront and chunk) wherever that gives every CU a workgroup ...
        const long t = ((M + 127) / 128) * ((Cout + 127) / 128);
        if (t >= 256) return {3, 1};
        // ... and where it does not (the 16 x 16 / 8 x 8 layers: few pixels, K = 9 Cin up to 4608), the same tile with the K range
        // cut over 2 / 4 / 8 workgroups and a fold pass: one 128 x 128 workgroup per CU runs at ~0.7 TFLOP/s, 64 x 64 tiles that
        // fill the machine at half that per flop (tools/conv_bench.py)
        const int nk = K / 16;
        if (t >= 32 && nk >= 128 && Cout % 4 == 0) {
            // every split keeps >= 64 chunks (a shorter main loop is dominated by its prologue / epilogue: the stride-2 layers,
            // K = 1152 / 2304, stay on 64 x 64 tiles), and the launch must reach ~one workgroup per CU
            int ks = 1;
            while (ks < 8 && t * ks < 256 && nk % (4 * ks) == 0 && nk / (2 * ks) >= 64) ks *= 2;
            if (ks > 1 && t * ks >= 192) return {3, ks};
        }
    }
    // 64 x 128 while that still gives >= 384 workgroups, else 64 x 64 (measured inside the step: filling the machine with the
    // smaller tile beats the larger tile on half the CUs, 39.8 vs 40.4 ms)
    return {t128 >= 384 ? 1 : 2, 1};
}
int tile_rows(int cfg) { return cfg == 0 ? Cfg256x64::BM : cfg == 3 ? Cfg128x128::BM : 64; }

}  // namespace

extern "C" int scp_conv_nhwc_partial_rows(int N, int H, int W, int Cin, int Cout, int ksize, int stride, int split, int* tiles_m,
                                          int* rows_per_tile) {
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long M = (long)N * Ho * Wo;
    const ConvPlan pl = plan_conv(M, Cout, ksize * ksize * Cin, split != 0);
    const int rows = pl.ksplit > 1 ? FOLD_ROWS : tile_rows(pl.cfg);
    if (tiles_m) *tiles_m = (int)((M + rows - 1) / rows);
    if (rows_per_tile) *rows_per_tile = rows;
    return 0;
}

extern "C" size_t scp_conv_nhwc_splitk_workspace(int N, int H, int W, int Cin, int Cout, int ksize, int stride, int split) {
    const int pad = ksize / 2;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const long M = (long)N * Ho * Wo;
    const ConvPlan pl = plan_conv(M, Cout, ksize * ksize * Cin, split != 0);
    return pl.ksplit > 1 ? (size_t)pl.ksplit * M * Cout * sizeof(float) : 0;
}

namespace {
int conv_forward_impl(const float* x, const float* w, const void* w_split, const float* bias, float* y, float* partials, unsigned* ticket,
                      const scp_bn::FwdFinalize* fin, int N, int H, int W, int Cin, int Cout, int ksize, int stride, int leaky,
                      float slope, void* splitk_ws, size_t splitk_bytes, void* stream, int scatter2 = 0) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return scp::fail(hipErrorInvalidValue, "conv_nhwc: empty problem");
    if (!x || (!w && !w_split) || !y || (leaky && !bias)) return scp::fail(hipErrorInvalidValue, "conv_nhwc: null argument");
    if (ksize != 1 && ksize != 3) return scp::fail(hipErrorInvalidValue, "conv_nhwc: kernel size must be 1 or 3");
    if (stride != 1 && stride != 2) return scp::fail(hipErrorInvalidValue, "conv_nhwc: stride must be 1 or 2");
    const int cpt = Cin / 16;
    if (Cin % 32 != 0 || (cpt & (cpt - 1))) return scp::fail(hipErrorInvalidValue, "conv_nhwc: Cin must be a power of two >= 32");
    if (fin && (Cout < 16 || Cout > 1024 || (Cout & (Cout - 1)))) return scp::fail(hipErrorInvalidValue, "conv_nhwc: BatchNorm statistics need a power-of-two Cout in [16,1024]");
    const int pad = ksize / 2;
    ConvArgs g{};
    g.x = x; g.w = w; g.w_split = w_split; g.bias = bias; g.y = y; g.partials = partials; g.ticket = ticket;
    g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.stride = stride; g.slope = slope; g.scatter2 = scatter2;
    g.Ho = (H + 2 * pad - ksize) / stride + 1;
    g.Wo = (W + 2 * pad - ksize) / stride + 1;
    const long M = (long)N * g.Ho * g.Wo, in_bytes = (long)N * H * W * Cin * 4;
    if (in_bytes >= (1l << 31) || M * Cout >= (1l << 31) || (long)Cout * ksize * ksize * Cin >= (1l << 30))
        return scp::fail(hipErrorInvalidValue, "conv_nhwc: tensor larger than 2^31 bytes");
    g.M = (int)M;
    g.K = ksize * ksize * Cin;
    g.x_bytes = (unsigned)in_bytes;
    g.lg_cpt = 0;
    while ((1 << g.lg_cpt) < cpt) g.lg_cpt++;
    if (fin) {
        g.fin = *fin;
        g.fin.R = M;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const ConvPlan pl = plan_conv(M, Cout, g.K, w_split != nullptr);
    const int cfg = pl.cfg;
    const bool stats = partials != nullptr;
    g.ksplit = pl.ksplit;
    if (pl.ksplit > 1) {
        // raw partial tiles by the convolution kernel, epilogue / statistics by the fold kernel
        if (!splitk_ws || splitk_bytes < (size_t)pl.ksplit * M * Cout * sizeof(float))
            return scp::fail(hipErrorInvalidValue, "conv_nhwc: this layer runs split-K, pass scp_conv_nhwc_splitk_workspace() bytes");
        g.ypart = static_cast<float*>(splitk_ws);
        g.nk_split = (g.K / 16) / pl.ksplit;
        if (ksize == 3) launch_cfg<Cfg128x128, 9>(g, false, false, st);
        else launch_cfg<Cfg128x128, 1>(g, false, false, st);
        g.tiles_m = (g.M + FOLD_ROWS - 1) / FOLD_ROWS;
        const dim3 grid(g.tiles_m * ((Cout + 255) / 256)), block(256);
        if (leaky) {
            if (stats) hipLaunchKernelGGL((conv_splitk_fold_kernel<EPI_BIAS_LEAKY, true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((conv_splitk_fold_kernel<EPI_BIAS_LEAKY, false>), grid, block, 0, st, g);
        } else {
            if (stats) hipLaunchKernelGGL((conv_splitk_fold_kernel<EPI_RAW, true>), grid, block, 0, st, g);
            else hipLaunchKernelGGL((conv_splitk_fold_kernel<EPI_RAW, false>), grid, block, 0, st, g);
        }
        return scp::check_launch("conv_nhwc_forward (split-K)");
    }
    if (ksize == 3) {
        if (cfg == 0) launch_cfg<Cfg256x64, 9>(g, leaky, stats, st);
        else if (cfg == 1) launch_cfg<Cfg64x128, 9>(g, leaky, stats, st);
        else if (cfg == 3) launch_cfg<Cfg128x128, 9>(g, leaky, stats, st);
        else launch_cfg<Cfg64x64, 9>(g, leaky, stats, st);
    } else {
        if (cfg == 0) launch_cfg<Cfg256x64, 1>(g, leaky, stats, st);
        else if (cfg == 1) launch_cfg<Cfg64x128, 1>(g, leaky, stats, st);
        else if (cfg == 3) launch_cfg<Cfg128x128, 1>(g, leaky, stats, st);
        else launch_cfg<Cfg64x64, 1>(g, leaky, stats, st);
    }
    return scp::check_launch("conv_nhwc_forward");
}
}  // namespace

extern "C" int scp_conv_nhwc_forward(const float* x, const float* w, const void* w_split, const float* bias, float* y, float* partials,
                                     int N, int H, int W, int Cin, int Cout, int ksize, int stride, int leaky, float slope,
                                     void* splitk_ws, size_t splitk_bytes, void* stream) {
    return conv_forward_impl(x, w, w_split, bias, y, partials, nullptr, nullptr, N, H, W, Cin, Cout, ksize, stride, leaky, slope, splitk_ws,
                             splitk_bytes, stream);
}

extern "C" int scp_conv_nhwc_forward_bn(const float* x, const float* w, const void* w_split, float* y, int N, int H, int W, int Cin, int Cout, int ksize,
                                        int stride, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                        long long* batches_tracked, float momentum, float eps, float* save_mean, float* save_invstd,
                                        float* save_scale, float* save_shift, void* workspace, size_t workspace_bytes, unsigned* ticket,
                                        void* splitk_ws, size_t splitk_bytes, void* stream) {
    if (!save_mean || !save_invstd || !save_scale || !save_shift || !workspace || !ticket)
        return scp::fail(hipErrorInvalidValue, "conv_nhwc_forward_bn: null argument");
    int tiles_m = 0;
    scp_conv_nhwc_partial_rows(N, H, W, Cin, Cout, ksize, stride, w_split != nullptr, &tiles_m, nullptr);
    if (workspace_bytes < (size_t)2 * tiles_m * Cout * sizeof(float)) return scp::fail(hipErrorInvalidValue, "conv_nhwc_forward_bn: workspace too small");
    const scp_bn::FwdFinalize fin{0, gamma, beta, running_mean, running_var, batches_tracked, momentum, eps, save_mean, save_invstd,
                                  save_scale, save_shift};
    return conv_forward_impl(x, w, w_split, nullptr, y, static_cast<float*>(workspace), ticket, &fin, N, H, W, Cin, Cout, ksize, stride, 0, 0.f,
                             splitk_ws, splitk_bytes, stream);
}

extern "C" int scp_conv1x1_nhwc_dgrad_stride2(const float* dy, const float* w_t, const void* w_t_split, float* dx, int N, int Ho, int Wo, int Cout,
                                              int Cin, void* stream) {
    if (!dy || (!w_t && !w_t_split) || !dx) return scp::fail(hipErrorInvalidValue, "conv1x1_nhwc_dgrad_stride2: null argument");
    if ((long)N * 4 * Ho * Wo * Cin >= (1l << 31)) return scp::fail(hipErrorInvalidValue, "conv1x1_nhwc_dgrad_stride2: tensor larger than 2^31 elements");
    // the 1x1 product at the dy resolution (dy as input, w_t = [Cin][Cout]), scattered by the epilogue
    return conv_forward_impl(dy, w_t, w_t_split, nullptr, dx, nullptr, nullptr, nullptr, N, Ho, Wo, Cout, Cin, 1, 1, 0, 0.f, nullptr, 0, stream, 1);
}

extern "C" int scp_conv_nhwc_dgrad_stride2(const float* dy, const void* w_dgrad_planes, float* dx, int N, int Ho, int Wo, int Cout, int Cin,
                                           void* stream) {
    if (!dy || !w_dgrad_planes || !dx) return scp::fail(hipErrorInvalidValue, "conv_nhwc_dgrad_stride2: null argument");
    if (N <= 0 || Ho <= 0 || Wo <= 0) return scp::fail(hipErrorInvalidValue, "conv_nhwc_dgrad_stride2: empty problem");
    const int cpt = Cout / 16;
    if (Cout % 32 != 0 || (cpt & (cpt - 1)) || Cin <= 0) return scp::fail(hipErrorInvalidValue, "conv_nhwc_dgrad_stride2: Cout must be a power of two >= 32");
    const long M = (long)N * Ho * Wo, dy_bytes = M * Cout * 4;
    if (dy_bytes >= (1l << 31) || 4 * M * Cin >= (1l << 31)) return scp::fail(hipErrorInvalidValue, "conv_nhwc_dgrad_stride2: tensor larger than 2^31 bytes");
    ConvArgs g{};
    g.x = dy; g.w_split = w_dgrad_planes; g.y = dx;
    g.H = Ho; g.W = Wo; g.Cin = Cout; g.Cout = Cin; g.M = (int)M; g.x_bytes = (unsigned)dy_bytes;
    while ((1 << g.lg_cpt) < cpt) g.lg_cpt++;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto cfg_tag) {
        using CFG = decltype(cfg_tag);
        g.nblk_n = (g.Cout + CFG::BN - 1) / CFG::BN;
        g.tiles_m = g.tiles_m_kernel = (g.M + CFG::BM - 1) / CFG::BM;
        const int total = g.ksplit * g.tiles_m * g.nblk_n;
        hipLaunchKernelGGL((conv_dgrad_s2_kernel<CFG>), dim3(((total + 7) >> 3) << 3), dim3(CFG::THREADS), 0, st, g);
    };
    // the largest tile that still gives ~one workgroup per CU with two workgroups per tile; tools/conv_bench.py (SCP_DGRAD2_PLAN=
    // "<tile 0..3><workgroups per tile 2|4>" overrides, for sweeps)
    auto wgs = [&](long bm, long bn) { return 2 * ((M + bm - 1) / bm) * ((Cin + bn - 1) / bn); };
    // measured (B = 32, 256^2: layer2.0 / 3.0 / 4.0 at 66 / 62 / 78 us, MIOpen 65 / 69 / 73): 256 x 64 tiles, one class per workgroup
    // for the 64-channel input; else two workgroups per tile while that fills the machine, and four 64 x 128 workgroups per tile
    // for the smallest maps
    int cfg = Cin <= 64 ? 0 : wgs(128, 128) >= 200 ? 3 : 1;
    g.ksplit = (Cin <= 64 || (cfg == 1 && wgs(64, 128) < 200)) ? 4 : 2;
    if (const char* e = getenv("SCP_DGRAD2_PLAN")) {
        if (e[0] >= '0' && e[0] <= '3' && (e[1] == '2' || e[1] == '4')) {
            if (Cin > 64 || e[0] == '0' || e[0] == '2') cfg = e[0] - '0';
            g.ksplit = e[1] - '0';
        }
    }
    if (cfg == 0) launch(Cfg256x64{});
    else if (cfg == 3) launch(Cfg128x128{});
    else if (cfg == 1) launch(Cfg64x128{});
    else launch(Cfg64x64{});
    return scp::check_launch("conv_nhwc_dgrad_stride2");
}

extern "C" int scp_conv_weight_planes_batch(const scp_conv_planes_desc* descs_device, int n, long long total_blocks, void* stream) {
    if (!descs_device || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL)
        return scp::fail(hipErrorInvalidValue, "conv_weight_planes_batch: bad argument");
    hipLaunchKernelGGL(conv_weight_planes_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       descs_device, n);
    return scp::check_launch("conv_weight_planes_batch");
}

extern "C" int scp_conv_weight_planes(const float* w, long long s_co, long long s_ci, long long s_ky, long long s_kx, int Cout, int Cin,
                                      int ksize, void* planes_fwd, void* planes_dgrad, void* stream) {
    if (!w || !planes_fwd || Cout <= 0 || Cin <= 0 || ksize <= 0) return scp::fail(hipErrorInvalidValue, "conv_weight_planes: bad argument");
    if ((ksize * ksize * Cin) % 16 || (planes_dgrad && (ksize * ksize * Cout) % 16))
        return scp::fail(hipErrorInvalidValue, "conv_weight_planes: k*k*Cin (and k*k*Cout for the input gradient) must be multiples of 16");
    const long n = (long)Cout * Cin * ksize * ksize;
    hipLaunchKernelGGL(conv_weight_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       (long)s_co, (long)s_ci, (long)s_ky, (long)s_kx, Cout, Cin, ksize, static_cast<__bf16*>(planes_fwd),
                       static_cast<__bf16*>(planes_dgrad));
    return scp::check_launch("conv_weight_planes");
}
