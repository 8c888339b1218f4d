// self-corr-pose_amd/csrc/gemm_core_split.h -- fp32 GEMM main loop on the bf16 matrix cores by EXACT operand splitting:
//     acc[M-tile][N-tile] += A[rows][K] * W[cols][K]^T,   A fp32, W given as three bf16 planes, accumulation in fp32.
//
// Every fp32 number x is the exact sum of three bf16 numbers h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): each
// residual is exactly representable in fp32 and has at most 16 resp. 8 significant bits left).  A product a*b is then the sum of
// nine bf16 x bf16 products, each EXACT in fp32; the six largest (hh, hm, mh, hl, lh, mm) are evaluated with
// v_mfma_f32_32x32x16_bf16 and accumulated in fp32, the three dropped ones (ml, lm, ll) are below 2^-24 |a b| each -- the size of
// the rounding of one fp32 fused multiply-add.  Measured against float64 the result is as close as the fp32-MFMA kernel's
// (tools/probes/gemm_split.hip prints both; tests/test_vit_gpu.py holds both to the same tolerance).
//
// Why: v_mfma_f32_32x32x2_f32 does 2 K-steps per 64 cycles, v_mfma_f32_32x32x16_bf16 16 K-steps per 32 cycles: six bf16 MFMAs
// per 16 K-steps cost 192 cycles against 512 for eight fp32 MFMAs -- a 2.67x higher ceiling for the same fp32-accurate result
// (dense bf16 peak / 6 = 417 TFLOP/s of fp32-equivalent work against 157 TFLOP/s).
//
// Layout.  The A tile is staged exactly as in gemm_core.h (LDS-DMA of fp32 rows, 64 B per row and chunk of 16 k, XOR-swizzled
// source); a lane's two ds_read_b128 of it are the 8 consecutive k of row (lane & 31) at k offset 8 (lane >> 5) -- the A operand
// layout of the 32x32x16 MFMA -- and are split in registers (VALU, in the shadow of the running MFMAs).  W is constant for the
// frozen ViT and pre-split once on the host side into planes [3][N][K] bf16; a chunk of a plane is 32 B per row (two 16-B slots,
// slot s of row r stored at s ^ ((r >> 3) & 1)), moved by the same LDS-DMA.
//
// Ring: two stages, one barrier per chunk -- top of chunk kc: vmcnt(0) [my pieces of chunk kc landed], s_barrier [everyone's
// did, and everyone has finished reading the other stage], issue the DMA of chunk kc+1 into the other stage, compute chunk kc.
// MFMAs, splits and fragment reads are left to the compiler's scheduler here (48 MFMAs of 32 cycles per wavefront and chunk
// against 14 ds_read_b128 and ~180 VALU instructions: nothing is tight), only the DMA and the synchronisation are asm.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "gemm_core.h"

namespace scp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// NPLANES = 3: the exact split (six products).  NPLANES = 1: operands ROUNDED to bf16 (W given as one bf16 plane, A converted in
// registers), one product -- plain bf16 matrix-core precision with fp32 accumulation, BASELINE configs[4] ("mixed bf16").
// ZSTART: the six partial products of a chunk are summed in a chunk-local accumulator that starts at zero and is added to the
// running accumulator by the VALU (see compute()); needs 16 WN more VGPRs, which the 4 x 2-tile wavefronts of the ViT's 256 x 128
// tile do not have while their A operand is still split in registers.
// APLANES (round 4): the A operand arrives PRE-SPLIT as three bf16 planes [3][rows][K] like W (written once by the epilogue of the
// kernel that produced it: csrc/vit_gemm.hip), moved by the same LDS-DMA -- the main loop is then DMA -> ds_read -> MFMA with no
// VALU split at all (the in-register split cost 5.4 VALU instructions per MFMA and was redone by every column block that read the
// same A rows: 9x for the qkv projection).
//
// TILED plane layout (APLANES only; "TP"): a [rows][K] operand is stored as  [rows / 32][K / 16][3 planes][32 rows][16 k]  bf16, i.e.
// every (32-row group, 16-k chunk, plane) is ONE contiguous KiB -- exactly what one wavefront's LDS-DMA instruction moves.  Measured
// (tools/probes/build_gemm_variants.sh, profiles/r04_gemm_planes_ablation.txt): with row-major planes a DMA instruction touches 32
// cache lines for its KiB (32 rows x 32 B) and the main loop is bound by that address traffic, not by the matrix pipe (removing 5/6
// of the MFMAs: -10 %; removing the DMA: -30 %); tiled, it touches 8.
// WTILED: the W planes are tiled (always with APLANES; a kernel that splits its fp32 A in registers may still take a tiled W).
template <int WM_, int WN_, int NWM_, int NWN_, int MINBLK_, int NPLANES_ = 3, bool ZSTART_ = (WM_ * WN_ <= 4), bool APLANES_ = false,
          bool WTILED_ = APLANES_>
struct SplitCfg {
    static constexpr int NPLANES = NPLANES_;
    static constexpr bool ZSTART = ZSTART_, APLANES = APLANES_, TILED = WTILED_;
    static_assert(!APLANES_ || WTILED_, "pre-split A comes with tiled W planes");
    static constexpr int WM = WM_, WN = WN_, NWM = NWM_, NWN = NWN_, NSTAGE = 2, MINBLK = MINBLK_;
    static constexpr int BM = 32 * WM * NWM, BN = 32 * WN * NWN, BK = 16;
    static constexpr int NW = NWM * NWN, THREADS = 64 * NW;
    static constexpr int A_GROUPS = BM / 32;
    static constexpr int A_PIECES = APLANES ? NPLANES * A_GROUPS : BM / 16;   // planes: 32 rows x 32 B; fp32: 16 rows x 64 B
    static constexpr int W_GROUPS = BN / 32, W_PIECES = NPLANES * W_GROUPS;   // 32 rows x 32 B per plane
    static_assert(APLANES || A_PIECES % NW == 0, "fp32 A pieces are dealt evenly to the wavefronts");
    // piece q goes to wavefront q % NW (its (q / NW)-th); a count that NW does not divide leaves some wavefronts one piece short
    static constexpr int A_PER = (A_PIECES + NW - 1) / NW, W_PER = (W_PIECES + NW - 1) / NW;
    static constexpr int A_PLANE_BYTES = BM * 32;
    static constexpr int A_BYTES = APLANES ? NPLANES * A_PLANE_BYTES : BM * 64, PLANE_BYTES = BN * 32;
    static constexpr int STAGE_BYTES = A_BYTES + NPLANES * PLANE_BYTES, LDS_BYTES = NSTAGE * STAGE_BYTES;
    static constexpr int NT = WM * WN;
};

struct Split3 { bf16x8 h, m, l; };
// element offset of (row, k) of plane p in the tiled plane layout of a [rows][K] operand (kchunks = K / 16)
__host__ __device__ __forceinline__ size_t tiled_plane_offset(int row, int k, int p, int kchunks) {
    return ((((size_t)(row >> 5) * kchunks + (k >> 4)) * 3 + p) << 9) + ((row & 31) << 4) + (k & 15);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
// x = h + m + l exactly, pairwise: 11 VALU instructions per two values (cvt, 2 unpack, 2 sub, cvt, 2 unpack, 2 sub, cvt)
__device__ __forceinline__ Split3 split3(f32x8 x) {
    u32x4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        h[p] = pack_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(h[p] << 16), r1 = x1 - __uint_as_float(h[p] & 0xffff0000u);
        m[p] = pack_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m[p] << 16), s1 = r1 - __uint_as_float(m[p] & 0xffff0000u);
        l[p] = pack_bf16(s0, s1);
    }
    Split3 s;
    s.h = __builtin_bit_cast(bf16x8, h);
    s.m = __builtin_bit_cast(bf16x8, m);
    s.l = __builtin_bit_cast(bf16x8, l);
    return s;
}

// A-tile source of the plain GEMM: A[M, K] fp32 row-major, one row per tile row (callers clamp rows to the matrix)
template <class CFG>
struct LinearASource {
    const char* a_base;
    unsigned a_off[CFG::A_PER];
    template <class FA>
    __device__ __forceinline__ void set_rows(const float* A, int K, int wave, int lane, FA a_row) {
        const int prow = lane >> 2, pslot = lane & 3, chunk = pslot ^ ((prow >> 2) & 3);
        a_base = reinterpret_cast<const char*>(A);
#pragma unroll
        for (int i = 0; i < CFG::A_PER; i++)
            a_off[i] = ((unsigned)a_row(16 * (wave * CFG::A_PER + i) + prow) * (unsigned)K + 4u * chunk) * 4u;
    }
    // piece I of this wavefront for chunk kc -> A part of the stage at `stage_lds`
    template <int I>
    __device__ __forceinline__ void issue(int kc, unsigned stage_lds, int wave) const {
        glds16(a_off[I], a_base + (size_t)kc * 64, stage_lds + (unsigned)(wave * CFG::A_PER + I) * 1024u);
    }
};

// A-tile source for pre-split operands: planes [NPLANES][rows_total][K] bf16; piece q = (plane q / A_GROUPS, 32 rows of group
// q % A_GROUPS), dealt to wavefront q % NW; slot s of row r lands at s ^ ((r >> 3) & 1) like W's
template <class CFG>
struct PlanesASource {
    const char* a_base;
    unsigned a_off[CFG::A_PER];
    template <class FA>
    __device__ __forceinline__ void set_rows(const void* A3, int rows_total, int K, int wave, int lane, FA a_row) {
        a_base = reinterpret_cast<const char*>(A3);
        const int prow = lane >> 1, pslot = lane & 1;
#pragma unroll
        for (int i = 0; i < CFG::A_PER; i++) {
            const int q = min(wave + CFG::NW * i, CFG::A_PIECES - 1), plane = q / CFG::A_GROUPS, r = 32 * (q % CFG::A_GROUPS) + prow;
            const int slot = pslot ^ ((r >> 3) & 1);
            a_off[i] = (unsigned)(tiled_plane_offset(a_row(r), 8 * slot, plane, K >> 4) * 2u);
        }
    }
    template <int I>
    __device__ __forceinline__ void issue(int kc, unsigned stage_lds, int wave) const {
        if ((I + 1) * CFG::NW <= CFG::A_PIECES || wave + CFG::NW * I < CFG::A_PIECES)          // wavefront-uniform
            glds16(a_off[I], a_base + (size_t)kc * 3072, stage_lds + (unsigned)(wave + CFG::NW * I) * 1024u);
    }
};

// an A source may also choose which chunk of W goes with chunk kc of A (`int w_chunk(int kc) const`: the stride-2 input gradient of
// csrc/conv_igemm.hip walks a subset of the nine taps of the weight planes); default: the same index
template <class T, class = void>
struct has_w_chunk : std::false_type {};
template <class T>
struct has_w_chunk<T, std::void_t<decltype(std::declval<const T&>().w_chunk(0))>> : std::true_type {};

template <class CFG, class ASRC = std::conditional_t<CFG::APLANES, PlanesASource<CFG>, LinearASource<CFG>>>
struct SplitGemmCore {
    struct Acc { f32x16 t[CFG::NT]; };

    ASRC asrc;
    const char* w_base;
    unsigned w_off[CFG::W_PER];
    char* lds;                     // generic pointer to the ring (compiler-visible reads)
    unsigned lds0;                 // its LDS byte address (DMA destinations)
    unsigned a_rd[2], w_rd;        // lane's fragment byte offsets inside a stage (tile 0)
    int wave, lane;
    int kc0 = 0;                   // first chunk of this workgroup's K range (split-K callers set it before run())

    __device__ __forceinline__ SplitGemmCore(float* lds_) {
        lds = reinterpret_cast<char*>(lds_);
        lds0 = SCP_LDS_ADDR(lds_);
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int half = lane >> 5, l31 = lane & 31;
        const int ra = row_base() + l31, rw = col_base() + l31;
        if constexpr (CFG::APLANES) {
            a_rd[0] = a_rd[1] = ra * 32 + 16 * (half ^ ((ra >> 3) & 1));
        } else {
#pragma unroll
            for (int c = 0; c < 2; c++) a_rd[c] = ra * 64 + 16 * ((2 * half + c) ^ ((ra >> 2) & 3));
        }
        w_rd = CFG::A_BYTES + rw * 32 + 16 * (half ^ ((rw >> 3) & 1));
    }
    __device__ __forceinline__ int row_base() const { return 32 * CFG::WM * (wave / CFG::NWN); }
    __device__ __forceinline__ int col_base() const { return 32 * CFG::WN * (wave % CFG::NWN); }

    // W3 = planes [3][N][K] bf16 (h, m, l); w_row maps a tile row (output column) to its source row
    template <class FW>
    __device__ __forceinline__ void set_w_rows(const void* W3, int N, int K, FW w_row) {
        w_base = reinterpret_cast<const char*>(W3);
        const int prow = lane >> 1, pslot = lane & 1;
#pragma unroll
        for (int i = 0; i < CFG::W_PER; i++) {
            const int q = min(wave + CFG::NW * i, CFG::W_PIECES - 1), plane = q / CFG::W_GROUPS, r = 32 * (q % CFG::W_GROUPS) + prow;
            const int slot = pslot ^ ((r >> 3) & 1);
            if constexpr (CFG::TILED) w_off[i] = (unsigned)(tiled_plane_offset(w_row(r), 8 * slot, plane, K >> 4) * 2u);
            else w_off[i] = ((unsigned)plane * (unsigned)N * (unsigned)K + (unsigned)w_row(r) * (unsigned)K + 8u * slot) * 2u;
        }
    }
    // the interface gemm_core.h's GemmCore shares (vit_gemm.hip picks a core per launch): A [M,K] fp32 row-major
    template <class FA, class FW>
    __device__ __forceinline__ void set_rows(const float* A, const void* W3, int N, int K, FA a_row, FW w_row) {
        static_assert(!CFG::APLANES, "planes A: use set_rows_planes");
        asrc.set_rows(A, K, wave, lane, a_row);
        set_w_rows(W3, N, K, w_row);
    }
    // pre-split A: planes [NPLANES][a_rows_total][K]
    template <class FA, class FW>
    __device__ __forceinline__ void set_rows_planes(const void* A3, int a_rows_total, const void* W3, int N, int K, FA a_row, FW w_row) {
        static_assert(CFG::APLANES, "fp32 A: use set_rows");
        asrc.set_rows(A3, a_rows_total, K, wave, lane, a_row);
        set_w_rows(W3, N, K, w_row);
    }
    __device__ __forceinline__ void set_linear_sources(const float* A, const void* W3, int m0, int n0, int M, int N, int K) {
        if constexpr (!CFG::APLANES) set_rows(A, W3, N, K, [&](int r) { return min(m0 + r, M - 1); }, [&](int r) { return min(n0 + r, N - 1); });
    }

    __device__ __forceinline__ void issue(int kc, int stage) const {
#ifdef SCP_PROBE_NO_DMA            // tools/probes: the main loop without its global -> LDS traffic (stale LDS contents)
        if (kc >= 2) return;
#endif
        const unsigned dst = lds0 + stage * CFG::STAGE_BYTES;
        kc += kc0;
        static_for<0, CFG::A_PER>([&](auto i) { asrc.template issue<decltype(i)::value>(kc, dst, wave); });
        int wkc = kc;
        if constexpr (has_w_chunk<ASRC>::value) wkc = __builtin_amdgcn_readfirstlane(asrc.w_chunk(kc));
        static_for<0, CFG::W_PER>([&](auto i) {
            constexpr int I = decltype(i)::value;
            if ((I + 1) * CFG::NW <= CFG::W_PIECES || wave + CFG::NW * I < CFG::W_PIECES)      // wavefront-uniform
                glds16(w_off[I], w_base + (size_t)wkc * (CFG::TILED ? 3072 : 32), dst + CFG::A_BYTES + (unsigned)(wave + CFG::NW * I) * 1024u);
        });
    }

    // chunk compute for pre-split A with zero-started chunk accumulators, software-pipelined at statement level: the A fragments of
    // row tile i + 1 are read while tile i's MFMAs run (two fragment sets); a tile's twelve MFMAs form two interleaved chains on
    // ITS pair of chunk accumulators (two pairs alternate), and the VALU folds a pair into the running accumulator while the NEXT
    // tile's MFMAs issue (sched_group_barrier: one MFMA, then two or three adds), so that the matrix pipe never waits for the adds;
    // only the last tile's fold of a chunk is exposed.  Without the fences the compiler keeps one chunk-accumulator pair per row
    // tile alive and spills.
    template <int S>
    __device__ __forceinline__ void compute_planes(Acc& acc) const {
        const char* st = lds + S * CFG::STAGE_BYTES;
        bf16x8 wf[3][CFG::WN];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int j = 0; j < CFG::WN; j++) wf[p][j] = *reinterpret_cast<const bf16x8*>(st + w_rd + p * CFG::PLANE_BYTES + j * 1024);
        bf16x8 af[2][3];
        auto load_a = [&](int i, bf16x8 (&f)[3]) {
#pragma unroll
            for (int p = 0; p < 3; p++) f[p] = *reinterpret_cast<const bf16x8*>(st + a_rd[0] + p * CFG::A_PLANE_BYTES + i * 1024);
        };
        load_a(0, af[0]);
        f32x16 c[2][CFG::WN];
        static_for<0, CFG::WM>([&](auto ii) {
            constexpr int i = decltype(ii)::value, cur = i & 1, prv = cur ^ 1;
            if constexpr (i + 1 < CFG::WM) load_a(i + 1, af[(i + 1) & 1]);
            const bf16x8 &ah = af[i & 1][0], &am = af[i & 1][1], &al = af[i & 1][2];
#pragma unroll
            for (int j = 0; j < CFG::WN; j++) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; r++) z[r] = 0.f;
                c[cur][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wf[1][j], z, 0, 0, 0);
            }
            auto mac = [&](const bf16x8& av, int p) {
#pragma unroll
                for (int j = 0; j < CFG::WN; j++) c[cur][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wf[p][j], c[cur][j], 0, 0, 0);
            };
#ifndef SCP_PROBE_NO_MFMA          // tools/probes: one MFMA per chain instead of six
            mac(al, 0);
            mac(ah, 2);
            mac(am, 0);
            mac(ah, 1);
#endif
            mac(ah, 0);
            if constexpr (i > 0) {
#pragma unroll
                for (int j = 0; j < CFG::WN; j++) {
                    acc.t[(i - 1) * CFG::WN + j] += c[prv][j];
                    asm volatile("" : "+v"(acc.t[(i - 1) * CFG::WN + j]));
                }
                // the ds_reads of the next tile's fragments first, then MFMA / VALU alternating (the previous tile's fold)
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int g_ = 0; g_ < 6 * CFG::WN; g_++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int j = 0; j < CFG::WN; j++) {
            acc.t[(CFG::WM - 1) * CFG::WN + j] += c[(CFG::WM - 1) & 1][j];
            asm volatile("" : "+v"(acc.t[(CFG::WM - 1) * CFG::WN + j]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    template <int S>
    __device__ __forceinline__ void compute(Acc& acc) const {
        if constexpr (CFG::APLANES && CFG::ZSTART && CFG::NPLANES == 3) {
            compute_planes<S>(acc);
            return;
        }
        const char* st = lds + S * CFG::STAGE_BYTES;
        bf16x8 wf[CFG::NPLANES][CFG::WN];
#pragma unroll
        for (int p = 0; p < CFG::NPLANES; p++)
#pragma unroll
            for (int j = 0; j < CFG::WN; j++) wf[p][j] = *reinterpret_cast<const bf16x8*>(st + w_rd + p * CFG::PLANE_BYTES + j * 1024);
#pragma unroll
        for (int i = 0; i < CFG::WM; i++) {
            Split3 a;
            if constexpr (CFG::APLANES) {
                a.h = *reinterpret_cast<const bf16x8*>(st + a_rd[0] + i * 1024);
                if constexpr (CFG::NPLANES == 3) {
                    a.m = *reinterpret_cast<const bf16x8*>(st + a_rd[0] + CFG::A_PLANE_BYTES + i * 1024);
                    a.l = *reinterpret_cast<const bf16x8*>(st + a_rd[0] + 2 * CFG::A_PLANE_BYTES + i * 1024);
                }
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(st + a_rd[0] + i * 2048);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(st + a_rd[1] + i * 2048);
                const f32x8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if constexpr (CFG::NPLANES == 1) a.h = __builtin_convertvector(x, bf16x8);
#ifdef SCP_PROBE_NO_SPLIT          // tools/probes: the A operand's VALU split replaced by one conversion (wrong values, timing only)
                else { a.h = __builtin_convertvector(x, bf16x8); a.m = a.h; a.l = a.h; }
#else
                else a = split3(x);
#endif
            }
            if constexpr (CFG::NPLANES == 1) {
#pragma unroll
                for (int j = 0; j < CFG::WN; j++)
                    acc.t[i * CFG::WN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, wf[0][j], acc.t[i * CFG::WN + j], 0, 0, 0);
                continue;
            }
            if constexpr (CFG::NPLANES == 3 && !CFG::ZSTART) {
                // smallest terms first; six products per accumulator tile, tiles interleaved so that consecutive MFMAs are independent
                auto mac = [&](const bf16x8& av, int p) {
#pragma unroll
                    for (int j = 0; j < CFG::WN; j++)
                        acc.t[i * CFG::WN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wf[p][j], acc.t[i * CFG::WN + j], 0, 0, 0);
                };
                mac(a.m, 1);
                mac(a.l, 0);
                mac(a.h, 2);
                mac(a.m, 0);
                mac(a.h, 1);
                mac(a.h, 0);
            }
            if constexpr (CFG::NPLANES == 3 && CFG::ZSTART) {
                // All six partial products of a chunk are summed by the matrix core in an accumulator that starts at ZERO in every
                // chunk (smallest terms first), and only the chunk's sum is added -- VALU, round to nearest -- to the running
                // accumulator.  Measured (tools/split_bias_probe.py, round 4): chained onto the running accumulator, every small-term
                // MFMA aligns its 16 products to the accumulator's exponent and FLOORS what falls below the matrix core's guard
                // bits: a relative bias of -1.1e-7 at K = 4608 on same-sign data (the fp32 cores: -4e-9), and the small terms were
                // rounded away one MFMA at a time.  Against a chunk-local accumulator nothing falls below the window.
                f32x16 c[CFG::WN];
#pragma unroll
                for (int j = 0; j < CFG::WN; j++) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; r++) z[r] = 0.f;
                    c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, wf[1][j], z, 0, 0, 0);
                }
                auto mac = [&](const bf16x8& av, int p) {
#pragma unroll
                    for (int j = 0; j < CFG::WN; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wf[p][j], c[j], 0, 0, 0);
                };
#ifndef SCP_PROBE_NO_MFMA
                mac(a.l, 0);
                mac(a.h, 2);
                mac(a.m, 0);
                mac(a.h, 1);
#endif
                mac(a.h, 0);
#pragma unroll
                for (int j = 0; j < CFG::WN; j++) {
                    acc.t[i * CFG::WN + j] += c[j];
                    asm volatile("" : "+v"(acc.t[i * CFG::WN + j]));      // pins the fold here: c[] dies before the next row tile
                }
                __builtin_amdgcn_sched_barrier(0);      // one row tile at a time: the chunk accumulators of two tiles never coexist
            }
        }
    }

    // acc = sum over nk chunks of 16 k (nk even and >= 2).  On return all LDS accesses and DMAs of this wavefront have completed.
    // (Round 5 tried a THREE-stage ring for the convolutions -- the DMA of chunk kc + 2 issued at the top of chunk kc: neutral in
    // isolation, -0.1 .. -0.2 ms inside the step, i.e. within noise, at 60 instead of 40 KiB of LDS; not kept.
    // profiles/r05_conv_ring_depth.txt.  Whoever retries it: wavefronts issue UNEQUAL numbers of DMA instructions per chunk when
    // the piece counts do not divide by the wavefront count, so the partial s_waitcnt has to be per wavefront.)
    __device__ __forceinline__ void run(Acc& acc, int nk) {
#pragma unroll
        for (int t = 0; t < CFG::NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc.t[t][r] = 0.f;
        issue(0, 0);
        for (int kc = 0; kc < nk; kc += 2) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            issue(kc + 1, 1);
            compute<0>(acc);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kc + 2 < nk) issue(kc + 2, 0);
            compute<1>(acc);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
};

}  // namespace scp
