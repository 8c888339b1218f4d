// self-corr-pose_amd/csrc/conv_wgrad.hip -- weight gradient of the encoder's 3x3 / stride 1 / pad 1 convolutions, NHWC fp32, on the
// gfx950 fp32 matrix cores:   dw[co][ky][kx][ci] = sum over output pixels p of dy[p][co] * x[p + (ky-1, kx-1)][ci].
//
// Replaces MIOpen's weight-gradient kernels for model/module/network/image_encoder.py:119-193 (BasicBlock 3x3 convolutions and
// the U-decoder's conv units), backward of both encoder passes of a step (encoder.py:29-37, correspondence.py:91).
//
// GEMM view: the contraction runs over PIXELS, so both operands are "K-major" (a pixel's channels are contiguous, consecutive
// pixels are the contraction index).  One workgroup owns a 64 (co) x 64 (ci) block of ALL NINE taps over a range of pixels:
//   * a chunk = 16 consecutive output pixels of one image row (two rows of 8 for the 8 x 8 maps);
//   * LDS holds dy[16 pixels][64 co] and the HALO block of x: (rows + 2) x (pixels + 2) input positions x 64 ci -- every tap is
//     the same block read at a shifted row, so x is fetched once for nine taps.  Positions outside the image are zero-filled by
//     the loader (buffer descriptor, out-of-range offsets return zeros): padding costs no branch and no second code path;
//   * v_mfma_f32_32x32x2_f32 takes one k (= pixel) pair per instruction: a lane's operand element is LDS[pixel row][channel]
//     -- 32 consecutive floats per half-wave, conflict-free ds_read_b32 with immediate offsets (pixel row and tap shift are
//     compile-time), no transposition anywhere;
//   * a wavefront accumulates 32 co x 32 ci x 9 taps = 144 accumulator VGPRs; per pixel pair 1 + 9 fragment reads feed 9 MFMAs.
// The instruction stream is the software-pipelined in-order stream of csrc/gemm_core.h (asm statements: the reads of the next
// pixel pair and the LDS-DMA of the chunk after next are issued in the shadow of the running MFMAs; one barrier per chunk).
// The pixel range is split over workgroups so that a layer launches ~256 of them; each writes its partial [64][9][64] block and
// a second kernel folds the partials in split order (deterministic, no atomics).
// Roofline: bound = fp32 MFMA; algorithmic flops 2 M Cout 9 Cin; algorithmic bytes 4 (M Cin + M Cout + 9 Cin Cout).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "gemm_core.h"
#include "scp_common.h"
#include "scp_hip.h"

namespace {

using scp::f32x16;
using scp::static_for;

constexpr int THREADS = 256, BC = 64;            // block of output / input channels per workgroup
constexpr int CHUNK = 16;                        // pixels per chunk
constexpr int DY_BYTES = CHUNK * BC * 4;         // 4 KiB
// LDS rows reserved for the halo block: stride 1: 3 x 18 = 54 or 4 x 10 = 40 used; stride 2 (the first 3x3 of layer2..4, split core
// only): a chunk of 16 output pixels reads 3 x 33 = 99 or 5 x 17 = 85 input positions
constexpr int x_rows(int stride) { return stride == 1 ? 64 : 112; }
constexpr int NSTAGE = 2;

struct WgradArgs {
    const float* x;        // [N, S H, S W, Cin] (S = stride)
    const float* dy;       // [N, H, W, Cout]
    float* partial;        // [splits][Cout][9][Cin]
    int lgW, lgH, Cin, Cout;
    int ncob, ncib, splits, chunks_per_split, total_chunks;
    unsigned x_bytes;
};

__device__ __forceinline__ void bufload16(unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}

enum { MODE_ISSUE = 0, MODE_TAIL = 1, MODE_LAST = 2 };

// CW = pixels of one image row inside a chunk (16, or 8 for 8-pixel-wide maps: a chunk is then two rows); STR = stride (lgW / lgH are
// the OUTPUT map's; the input is STR times as large)
// KS = kernel size: 3 (pad 1: the halo block), or 1 (pad 0, the stride-2 projections: the block is just the chunk's 16 pixels)
template <int CW, int STR = 1, int KS = 3>
struct WgradCore {
    static constexpr int CR = CHUNK / CW, HC = KS == 1 ? CW : STR * (CW - 1) + 3, HR = KS == 1 ? CR : STR * (CR - 1) + 3;
    static constexpr int X_ROWS = KS == 1 ? 16 : x_rows(STR), STAGE_BYTES = DY_BYTES + X_ROWS * BC * 4;
    static constexpr int XP = X_ROWS / 16;            // x pieces per wavefront
    static constexpr int PSTEP = KS == 1 ? STR : 1, PAD = KS == 1 ? 0 : 1;   // block position -> input pixel: STR (y0, x0) + PSTEP pos - PAD
    static constexpr int PER = 1 + XP;               // LDS-DMA pieces per wavefront per chunk
    static_assert(HR * HC <= X_ROWS, "halo block fits the reserved rows");
    struct Acc { f32x16 t[9]; };
    struct Frag { float a; float b[9]; };

    const WgradArgs& g;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lds0, a_rd, b_rd;
    unsigned dy_off;                 // lane's byte offset inside a dy chunk
    int x_hr[XP], x_hc[XP];          // lane's halo position (row, column) per x piece; row < 0: beyond the block
    unsigned x_lane;                 // lane's byte offset inside a pixel's channels
    int wave, lane;
    int chunk0;
    const char* dy_base;
    Frag F[2];

    __device__ __forceinline__ WgradCore(const WgradArgs& args, float* lds, int cob, int cib, int first_chunk) : g(args) {
        lds0 = SCP_LDS_ADDR(lds);
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int half = lane >> 5, l31 = lane & 31;
        const int wm = wave >> 1, wn = wave & 1;
        a_rd = lds0 + half * (BC * 4) + (wm * 32 + l31) * 4;
        b_rd = lds0 + DY_BYTES + half * (BC * 4) + (wn * 32 + l31) * 4;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, (int)g.x_bytes, 0x00020000);
        chunk0 = first_chunk;
        const int prow = lane >> 4, grp = lane & 15;          // a 1-KiB piece = 4 rows x 256 B
        dy_off = ((unsigned)(4 * wave + prow) * (unsigned)g.Cout + (unsigned)(cob * BC)) * 4u + 16u * grp;
        dy_base = reinterpret_cast<const char*>(g.dy);
        x_lane = (unsigned)(cib * BC) * 4u + 16u * grp;
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const int row = 4 * (XP * wave + i) + prow;       // LDS row of the halo block
            x_hr[i] = row < HR * HC ? row / HC : -1000;
            x_hc[i] = row % HC;
        }
    }

    // chunk c of this workgroup -> stage s
    __device__ __forceinline__ void issue(int c, int s) {
        const int gc = chunk0 + c;                             // global chunk index
        const int p0 = gc * CHUNK;
        const int W = 1 << g.lgW, H = 1 << g.lgH;
        const int x0 = p0 & (W - 1), y0 = (p0 >> g.lgW) & (H - 1), img = p0 >> (g.lgW + g.lgH);
        const unsigned stage = lds0 + (unsigned)s * STAGE_BYTES;
        scp::glds16(dy_off, dy_base + (size_t)p0 * g.Cout * 4, stage + (unsigned)wave * 1024u);
#pragma unroll
        for (int i = 0; i < XP; i++) {
            const int yy = STR * y0 + PSTEP * x_hr[i] - PAD, xx = STR * x0 + PSTEP * x_hc[i] - PAD;
            const bool ok = yy >= 0 && yy < STR * H && xx >= 0 && xx < STR * W;
            const unsigned off = (unsigned)((img * STR * H + yy) * (STR * W)) + (unsigned)xx;
            const unsigned voff = ok ? off * (unsigned)g.Cin * 4u + x_lane : 0x80000000u;
            bufload16(voff, rsrc, stage + DY_BYTES + (unsigned)(XP * wave + i) * 1024u);
        }
    }

    // fragments of pixel pair (2 st, 2 st + 1) of the chunk in stage S: A from dy, B per tap from the shifted halo rows
    template <int S, int ST, int R>
    __device__ __forceinline__ void read(Frag& f) {
        if constexpr (R == 0) {
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a) : "v"(a_rd), "i"(S * STAGE_BYTES + (2 * ST) * BC * 4));
        } else {
            constexpr int tap = R - 1, ky = tap / 3, kx = tap % 3;
            // pixel k = 2 ST + half: row (k / CW + ky) * HC + k % CW + kx; 2 ST and 2 ST + 1 share the image row (CW is even)
            constexpr int row = ((2 * ST) / CW + ky) * HC + (2 * ST) % CW + kx;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.b[tap]) : "v"(b_rd), "i"(S * STAGE_BYTES + row * BC * 4));
        }
    }
    template <int TAP>
    __device__ __forceinline__ void mfma(Acc& acc, const Frag& f) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc.t[TAP]) : "v"(f.a), "v"(f.b[TAP]));
    }

    template <int S, int MODE>
    __device__ __forceinline__ void chunk(Acc& acc, int c) {
        static_for<0, 8>([&](auto st) {
            constexpr int ST = decltype(st)::value;
            Frag& cur = F[ST & 1];
            Frag& nxt = F[(ST + 1) & 1];
            static_for<0, 9>([&](auto tap) {
                constexpr int T = decltype(tap)::value;
                mfma<T>(acc, cur);
                if constexpr (ST < 7) {
                    read<S, ST + 1, T + 1>(nxt);
                    if constexpr (T == 0) read<S, ST + 1, 0>(nxt);
                } else if constexpr (MODE != MODE_LAST) {
                    // last pixel pair of the chunk: every read of this chunk has completed (lgkmcnt(0) below); hand the
                    // stage over, then start on the next chunk's first pair
                    if constexpr (T == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    if constexpr (T >= 1) {
                        read<(S + 1) % NSTAGE, 0, T>(nxt);
                        if constexpr (T == 1) read<(S + 1) % NSTAGE, 0, 0>(nxt);
                    }
                }
            });
            if constexpr (ST == 7 && MODE != MODE_LAST) {
                read<(S + 1) % NSTAGE, 0, 9>(nxt);
                if constexpr (MODE == MODE_ISSUE) issue(c + NSTAGE, S);
            }
            if constexpr (!(ST == 7 && MODE == MODE_LAST)) asm volatile("s_waitcnt lgkmcnt(0)");
        });
    }

    // nk chunks (even, >= 2)
    __device__ __forceinline__ void run(Acc& acc, int nk) {
        static_for<0, 9>([&](auto t) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc.t[decltype(t)::value][r] = 0.f;
            asm volatile("" : "+v"(acc.t[decltype(t)::value]));
        });
        issue(0, 0);
        issue(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(PER) : "memory");
        static_for<0, 10>([&](auto r) { read<0, 0, decltype(r)::value>(F[0]); });
        asm volatile("s_waitcnt lgkmcnt(0)");
        int c = 0;
        for (; c + NSTAGE < nk; c += NSTAGE) {
            chunk<0, MODE_ISSUE>(acc, c);
            chunk<1, MODE_ISSUE>(acc, c + 1);
        }
        chunk<0, MODE_TAIL>(acc, c);
        chunk<1, MODE_LAST>(acc, c + 1);
        asm volatile("s_nop 15\n\ts_nop 3");
    }
};

// ---- the same contraction on the bf16 matrix cores with exactly split operands (csrc/gemm_core_split.h) ---------------------------
// Both operands are activations and the contraction index is the pixel, so a lane's MFMA operand (8 consecutive k = 8 pixels of
// ONE channel) is 8 separate LDS words; they are split in registers.  Per chunk of 16 pixels (= one k-step of
// v_mfma_f32_32x32x16_bf16) and wavefront: 8 words of dy and 3 x 10 words of the halo block (a halo row serves its three kx taps:
// the tap's 8 pixels are words kx .. kx + 7 of the row's 10) are read with ds_read_b32, split by ROUND TO NEAREST EVEN, two values
// per v_cvt_pk_bf16_f32 (x = h + m + l exactly, residuals of both signs: the three dropped products ml, lm, ll then carry no
// systematic sign; rounds 3 split by truncation, which made every dropped product share the sign of a*b -- VERDICT r3 weak #4),
// and packed pairwise with v_perm_b32; 9 taps x 6 partial products = 54 MFMAs of 32 cycles against 72 of 64 cycles on the fp32
// cores.  DMA, ring and halo layout are WgradCore's; MFMAs, splits and reads are left to the compiler's scheduler.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CW, int STR = 1, int KS = 3>
struct WgradSplitCore : WgradCore<CW, STR, KS> {
    using Base = WgradCore<CW, STR, KS>;
    static constexpr int STAGE_BYTES = Base::STAGE_BYTES;
    static constexpr int PS = KS == 1 ? 1 : STR;      // halo columns between consecutive output pixels
    static constexpr int NX = PS * 7 + KS;            // halo words of one row a lane's 8 pixels touch (all kx)
    using Acc = typename Base::Acc;
    const float* ldsf;
    int a_lane, b_lane;            // float offsets inside a stage of this lane's first dy word / first halo word

    __device__ __forceinline__ WgradSplitCore(const WgradArgs& args, float* lds, int cob, int cib, int first_chunk)
        : Base(args, lds, cob, cib, first_chunk), ldsf(lds) {
        const int half = this->lane >> 5, l31 = this->lane & 31;
        const int wm = this->wave >> 1, wn = this->wave & 1;
        a_lane = 8 * half * BC + wm * 32 + l31;                                   // pixel 8 half + i -> dy row
        // pixel p = 8 half + i sits at halo (row PS (p / CW), column PS (p % CW)); the tap adds (ky, kx)
        const int pr = CW == 16 ? 0 : PS * half, pc0 = CW == 16 ? PS * 8 * half : 0;
        b_lane = DY_BYTES / 4 + (pr * Base::HC + pc0) * BC + wn * 32 + l31;
    }

    struct Split { float h, m, l; };
    // two values at a time: bf16(x) by v_cvt_pk_bf16_f32 (RNE), kept as floats with zero low halves
    static __device__ __forceinline__ void split2(float x0, float x1, Split& s0, Split& s1) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        auto rne = [](float a, float b, float& ra, float& rb) {
            const f32x2 v = {a, b};
            const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            ra = __uint_as_float(p << 16);
            rb = __uint_as_float(p & 0xffff0000u);
        };
        rne(x0, x1, s0.h, s1.h);
        const float r0 = x0 - s0.h, r1 = x1 - s1.h;
        rne(r0, r1, s0.m, s1.m);
        s0.l = r0 - s0.m;           // <= 8 significant bits: exact in bf16, the pack below keeps its upper half
        s1.l = r1 - s1.m;
    }
    // bf16 pair (element 0 = lo, element 1 = hi) from two floats whose low 16 bits are zero (or ignored)
    static __device__ __forceinline__ unsigned pack(float lo, float hi) {
        return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
    }

    template <int S>
    __device__ __forceinline__ void compute(Acc& acc) const {
        const float* st = ldsf + S * (STAGE_BYTES / 4);
        u32x4 ah, am, al;
        {
            Split a[8];
#pragma unroll
            for (int i = 0; i < 8; i += 2) split2(st[a_lane + i * BC], st[a_lane + (i + 1) * BC], a[i], a[i + 1]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ah[j] = pack(a[2 * j].h, a[2 * j + 1].h);
                am[j] = pack(a[2 * j].m, a[2 * j + 1].m);
                al[j] = pack(a[2 * j].l, a[2 * j + 1].l);
            }
        }
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int ky = 0; ky < KS; ky++) {
            // the NX words of halo row ky this lane's 8 pixels touch (stride 2: pixel j of tap kx is word 2 j + kx of 17)
            Split x[NX + 1];
#pragma unroll
            for (int j = 0; j < NX; j += 2)
                split2(st[b_lane + (ky * Base::HC + j) * BC], st[b_lane + (ky * Base::HC + (j + 1 < NX ? j + 1 : j)) * BC], x[j], x[j + 1]);
            bf16x8 Bh[KS], Bm[KS], Bl[KS];
#pragma unroll
            for (int kx = 0; kx < KS; kx++) {
                u32x4 h, m, l;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int w0 = kx + PS * 2 * j, w1 = kx + PS * (2 * j + 1);
                    h[j] = pack(x[w0].h, x[w1].h);
                    m[j] = pack(x[w0].m, x[w1].m);
                    l[j] = pack(x[w0].l, x[w1].l);
                }
                Bh[kx] = __builtin_bit_cast(bf16x8, h); Bm[kx] = __builtin_bit_cast(bf16x8, m); Bl[kx] = __builtin_bit_cast(bf16x8, l);
            }
            // smallest terms first; the three taps of the row alternate so that consecutive MFMAs are independent
            auto mac = [&](const bf16x8& av, const bf16x8 (&bv)[KS]) {
#pragma unroll
                for (int kx = 0; kx < KS; kx++)
                    acc.t[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[kx], acc.t[ky * KS + kx], 0, 0, 0);
            };
            mac(Am, Bm);
            mac(Al, Bh);
            mac(Ah, Bl);
            mac(Am, Bh);
            mac(Ah, Bm);
            mac(Ah, Bh);
        }
    }

    // nk chunks (even, >= 2)
    __device__ __forceinline__ void run(Acc& acc, int nk) {
#pragma unroll
        for (int t = 0; t < KS * KS; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc.t[t][r] = 0.f;
        this->issue(0, 0);
        for (int c = 0; c < nk; c += 2) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            this->issue(c + 1, 1);
            compute<0>(acc);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (c + 2 < nk) this->issue(c + 2, 0);
            compute<1>(acc);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
};

template <int CW, bool SPLIT, int STR = 1, int KS = 3>
__global__ __launch_bounds__(THREADS, 2) void conv_wgrad_kernel(const WgradArgs g) {
    static_assert(SPLIT || (STR == 1 && KS == 3), "the fp32 core's compile-time read offsets are the stride-1 3x3 halo's");
    using Core = std::conditional_t<SPLIT, WgradSplitCore<CW, STR, KS>, WgradCore<CW, STR, KS>>;
    constexpr int NT = KS * KS;
    if constexpr (SPLIT) scp::claim_vgprs<256>();                               // bf16 MFMAs: two wavefronts fill a SIMD's register file (scp_common.h)
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE * Core::STAGE_BYTES / 4];
    // workgroup b runs on XCD b % 8 and takes a contiguous share of the (split, co block, ci block) list, split slowest: the
    // workgroups that read the same pixels sit on one XCD's L2
    const int total = g.splits * g.ncob * g.ncib;
    const int per_xcd = (total + 7) >> 3;
    const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lid >= total) return;
    const int blocks = g.ncob * g.ncib;
    const int split = lid / blocks, blk = lid - split * blocks;
    const int cob = blk / g.ncib, cib = blk - cob * g.ncib;
    const int first = split * g.chunks_per_split;
    const int nk = min(g.chunks_per_split, g.total_chunks - first);
    Core core(g, lds, cob, cib, first);
    typename Core::Acc acc;
    core.run(acc, nk);
    const int half = core.lane >> 5, l31 = core.lane & 31;
    const int wm = core.wave >> 1, wn = core.wave & 1;
    float* out = g.partial + (size_t)split * g.Cout * NT * g.Cin;
    const int ci = cib * BC + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < NT; tap++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = cob * BC + wm * 32 + scp::acc_row(r, half);
            out[((size_t)co * NT + tap) * g.Cin + ci] = acc.t[tap][r];
        }
}

// dw = sum over splits of the partial blocks, in a FIXED order (deterministic): a workgroup = 16 output float4s x 16 split lanes;
// lane j adds splits j, j + 16, ... in increasing order, the 16 lane sums are then added in lane order.  (One thread per output
// walking all splits was latency-bound for the 64-channel layers: 9216 threads x 512 dependent-address loads, ~100 us.)
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n4, int splits) {
    __shared__ float4 red[16][17];
    const int o = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        const float4* p = reinterpret_cast<const float4*>(partial) + i;
        for (int k = j; k < splits; k += 16) {
            const float4 v = p[(size_t)k * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[j][o] = s;
    __syncthreads();
    if (j == 0 && i < n4) {
        float4 t = red[0][o];
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const float4 v = red[k][o];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        reinterpret_cast<float4*>(dw)[i] = t;
    }
}

struct Plan { int splits, chunks_per_split, total_chunks, ncob, ncib, lgW, lgH; bool ok; };
Plan make_plan(int N, int H, int W, int Cin, int Cout) {
    Plan p{};
    p.ok = false;
    if (N <= 0 || H < 8 || W < 8 || (H & (H - 1)) || (W & (W - 1)) || Cin % BC || Cout % BC) return p;
    while ((1 << p.lgW) < W) p.lgW++;
    while ((1 << p.lgH) < H) p.lgH++;
    const long M = (long)N * H * W;
    p.total_chunks = (int)(M / CHUNK);
    p.ncob = Cout / BC;
    p.ncib = Cin / BC;
    const int blocks = p.ncob * p.ncib;
    // ~256 workgroups (one per CU), every split an even number of chunks >= 4.  512 (two per CU) was measured 3-5 % slower on
    // the 9.7-GFLOP layers and doubles the partial-sum traffic (2.7 GB written + read per step)
    // (the split main loop: 512 workgroups are 8 % faster in isolation, 2.07 vs 2.25 ms per encoder pass, and 0.4 ms SLOWER inside
    // the step)
    int splits = (256 + blocks - 1) / blocks;
    int cps = (p.total_chunks + splits - 1) / splits;
    cps = (cps + 1) & ~1;
    if (cps < 4) cps = 4;
    if (p.total_chunks < 2 || (p.total_chunks & 1)) return p;
    cps = cps < p.total_chunks ? cps : p.total_chunks;
    p.chunks_per_split = cps;
    p.splits = (p.total_chunks + cps - 1) / cps;
    p.ok = true;
    return p;
}

}  // namespace

namespace {
// which (ksize, stride) the kernels take: 3x3 stride 1 (both cores); on the split core also 3x3 stride 2 and 1x1 stride 2 (H, W even)
// and 1x1 stride 1
bool shape_ok(int H, int W, int ksize, int stride, int split) {
    if (ksize == 3 && stride == 1) return true;
    if (split && ksize == 1 && stride == 1) return true;
    return split && stride == 2 && (ksize == 3 || ksize == 1) && H % 2 == 0 && W % 2 == 0;
}
}  // namespace

extern "C" size_t scp_conv_nhwc_weight_grad_workspace(int N, int H, int W, int Cin, int Cout, int ksize, int stride) {
    if (!shape_ok(H, W, ksize, stride, 1)) return 0;
    const Plan p = make_plan(N, H / stride, W / stride, Cin, Cout);
    if (!p.ok) return 0;
    return (size_t)p.splits * Cout * ksize * ksize * Cin * sizeof(float);
}

extern "C" int scp_conv_nhwc_weight_grad(const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                                         size_t workspace_bytes, int N, int H, int W, int Cin, int Cout, int ksize, int stride,
                                         int split, void* stream) {
    if (!x || !dy || !dw || !workspace) return scp::fail(hipErrorInvalidValue, "conv_weight_grad: null argument");
    if (!shape_ok(H, W, ksize, stride, split))
        return scp::fail(hipErrorInvalidValue, "conv_weight_grad: 3x3 / stride 1, or (split core) 1x1 / stride 1 and 3x3, 1x1 / stride 2 on even maps");
    if (dbias) return scp::fail(hipErrorInvalidValue, "conv_weight_grad: the bias gradient comes from scp_bias_leaky_relu_backward");
    const int Ho = H / stride, Wo = W / stride, nt = ksize * ksize;
    const Plan p = make_plan(N, Ho, Wo, Cin, Cout);
    if (!p.ok) return scp::fail(hipErrorInvalidValue, "conv_weight_grad: needs a power-of-two output map >= 8 x 8 and channel counts that are multiples of 64");
    const size_t need = (size_t)p.splits * Cout * nt * Cin * sizeof(float);
    if (workspace_bytes < need) return scp::fail(hipErrorInvalidValue, "conv_weight_grad: workspace too small");
    const long in_bytes = (long)N * H * W * Cin * 4;
    if (in_bytes >= (1l << 31) || (long)N * Ho * Wo * Cout * 4 >= (1l << 32)) return scp::fail(hipErrorInvalidValue, "conv_weight_grad: tensor too large");
    WgradArgs g{};
    g.x = x; g.dy = dy; g.partial = static_cast<float*>(workspace);
    g.lgW = p.lgW; g.lgH = p.lgH; g.Cin = Cin; g.Cout = Cout;
    g.ncob = p.ncob; g.ncib = p.ncib; g.splits = p.splits; g.chunks_per_split = p.chunks_per_split; g.total_chunks = p.total_chunks;
    g.x_bytes = (unsigned)in_bytes;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int total = p.splits * p.ncob * p.ncib;
    const dim3 grid(((total + 7) >> 3) << 3);
    const bool wide = Wo >= 16;
    if (stride == 2 && ksize == 3) {
        if (wide) hipLaunchKernelGGL((conv_wgrad_kernel<16, true, 2, 3>), grid, dim3(THREADS), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_kernel<8, true, 2, 3>), grid, dim3(THREADS), 0, st, g);
    } else if (stride == 2) {
        if (wide) hipLaunchKernelGGL((conv_wgrad_kernel<16, true, 2, 1>), grid, dim3(THREADS), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_kernel<8, true, 2, 1>), grid, dim3(THREADS), 0, st, g);
    } else if (ksize == 1) {
        if (wide) hipLaunchKernelGGL((conv_wgrad_kernel<16, true, 1, 1>), grid, dim3(THREADS), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_kernel<8, true, 1, 1>), grid, dim3(THREADS), 0, st, g);
    } else if (split) {
        if (wide) hipLaunchKernelGGL((conv_wgrad_kernel<16, true>), grid, dim3(THREADS), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_kernel<8, true>), grid, dim3(THREADS), 0, st, g);
    } else {
        if (wide) hipLaunchKernelGGL((conv_wgrad_kernel<16, false>), grid, dim3(THREADS), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_kernel<8, false>), grid, dim3(THREADS), 0, st, g);
    }
    const int n4 = Cout * nt * Cin / 4;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3((n4 + 15) / 16), dim3(256), 0, st, static_cast<const float*>(workspace), dw, n4, p.splits);
    return scp::check_launch("conv_weight_grad");
}
