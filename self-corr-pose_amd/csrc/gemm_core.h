// self-corr-pose_amd/csrc/gemm_core.h -- the fp32 matrix-core main loop shared by the ViT linear layers (csrc/vit_gemm.hip)
// and the encoder's implicit-GEMM convolutions (csrc/conv_igemm.hip):  acc[M-tile][N-tile] += A[rows][K] * W[cols][K]^T
// with both operands K-contiguous ("row r, 16 consecutive k" = 64 bytes = one LDS-DMA row).
//
// What it replaces in the reference: nothing by itself -- it is the contraction engine under
//   third-party/zsp/zsp/method/vision_transformer_flexible.py:54-70,85-101   (Linear layers of the ViT)
//   model/module/network/image_encoder.py:119-193                            (3x3 / 1x1 convolutions of the encoder)
//
// CDNA4 mapping.  v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate, 64 cycles per instruction per SIMD).  A
// workgroup of NWM x NWN wavefronts owns a (32 WM NWM) x (32 WN NWN) tile; a wavefront owns WM x WN MFMA tiles (16 accumulator
// VGPRs each).  K advances in chunks of 16: a chunk of an operand tile is rows x 64 B, moved global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: 16 rows x 64 B = 1 KiB per wavefront instruction, no staging VGPRs) into an NSTAGE-deep ring.
// The DMA destination is lane-linear, so LDS bank conflicts are removed by permuting the SOURCE: 16-byte slot c of tile row r
// is filled from chunk c ^ ((r >> 2) & 3) and read back through the same XOR.
//
// The main loop is ONE in-order instruction stream per wavefront, written as a sequence of `asm volatile` statements (hipcc
// keeps their relative order): every K chunk is two halves of 4 WM WN MFMAs; while a half's MFMAs issue, the fragment reads
// (ds_read_b128) of the NEXT half and the LDS-DMA of the chunk NSTAGE ahead are issued in the shadow of the running MFMA, so
// that no wavefront ever waits for LDS or global latency with the matrix pipe idle (round 2's kernel issued all reads of a
// chunk, waited, and then issued its MFMAs, relying on 3 wavefronts per SIMD to fill the gaps: 0.69 of peak in isolation).
// One s_barrier per chunk, placed two MFMAs into the second half.
//
// Ring protocol per wavefront (PER = LDS-DMA pieces per wavefront per chunk):
//   prologue: issue D(0..NSTAGE-1); vmcnt(PER (NSTAGE-1)); barrier; R0(0); lgkmcnt(0)
//   chunk kc: H0: MFMAs on F0, reads R1(kc) -> F1; lgkmcnt(0)                       [all my reads of chunk kc are complete]
//             H1: MFMA, MFMA, vmcnt(PER (NSTAGE-2)) [my pieces of D(kc+1) landed], s_barrier [everyone's did, and everyone is
//                 done with stage kc % NSTAGE], then MFMAs on F1 interleaved with D(kc+NSTAGE) into the freed stage and the
//                 reads R0(kc+1) -> F0; lgkmcnt(0)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace scp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SCP_LDS_ADDR(p) ((unsigned)(size_t)((__attribute__((address_space(3))) void*)(p)))

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int WM_, int WN_, int NWM_, int NWN_, int NSTAGE_, int MINBLK_>
struct GemmCfg {
    static constexpr int WM = WM_, WN = WN_, NWM = NWM_, NWN = NWN_, NSTAGE = NSTAGE_, MINBLK = MINBLK_;
    static constexpr bool APLANES = false;                                        // (csrc/gemm_core_split.h: pre-split A operand)
    static constexpr int BM = 32 * WM * NWM, BN = 32 * WN * NWN, BK = 16;
    static constexpr int NW = NWM * NWN, THREADS = 64 * NW;
    static constexpr int A_PIECES = BM / 16, W_PIECES = BN / 16;                  // 1-KiB LDS-DMA instructions per chunk
    static constexpr int A_PER = A_PIECES / NW, W_PER = W_PIECES / NW, PER = A_PER + W_PER;
    static_assert(A_PIECES % NW == 0 && W_PIECES % NW == 0, "pieces are dealt evenly to the wavefronts");
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 4, W_BASE_BYTES = BM * BK * 4;
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
    static constexpr int NT = WM * WN;                                            // accumulator tiles per wavefront
    static constexpr int NM = NT * 4;                                             // MFMAs per half chunk
    static constexpr int NREAD = WM + WN;                                         // ds_read_b128 per half chunk
    static_assert((NSTAGE - 1) * STAGE_BYTES + W_BASE_BYTES / 1 < (1 << 30), "");
};

template <int N>
using ic = std::integral_constant<int, N>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

// one LDS-DMA piece: lanes' 16-byte sources (sbase + voff) -> LDS bytes [lds_dst, lds_dst + 1024), lane-linear.  M0 carries
// the LDS destination and is written in the same statement that reads it (the compiler owns M0 between statements).
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

enum { GEMM_STEP_ISSUE = 0, GEMM_STEP_TAIL = 1, GEMM_STEP_LAST = 2 };

// Source policy of the plain GEMM: A[M, K] and W[N, K] row-major, rows clamped to the matrix (callers mask their stores).
template <class CFG>
struct LinearSource {
    const char* a_base;            // A + k offset of chunk 0
    const char* w_base;
    unsigned a_off[CFG::A_PER > 0 ? CFG::A_PER : 1], w_off[CFG::W_PER > 0 ? CFG::W_PER : 1];   // per lane: byte offset of its 16 B
    // a_row(r) / w_row(r): source row of tile row r (callers clamp to the matrix and may gather through an index list)
    template <class FA, class FW>
    __device__ __forceinline__ void set_rows(const float* A, const float* W, int K, int wave, int lane, FA a_row, FW w_row) {
        const int prow = lane >> 2, pslot = lane & 3;
        const int chunk = pslot ^ ((prow >> 2) & 3);
        a_base = reinterpret_cast<const char*>(A);
        w_base = reinterpret_cast<const char*>(W);
#pragma unroll
        for (int i = 0; i < CFG::A_PER; i++) a_off[i] = ((unsigned)a_row(16 * (wave * CFG::A_PER + i) + prow) * (unsigned)K + 4u * chunk) * 4u;
#pragma unroll
        for (int i = 0; i < CFG::W_PER; i++) w_off[i] = ((unsigned)w_row(16 * (wave * CFG::W_PER + i) + prow) * (unsigned)K + 4u * chunk) * 4u;
    }
    __device__ __forceinline__ void set(const float* A, const float* W, int m0, int n0, int M, int N, int K, int wave, int lane) {
        set_rows(A, W, K, wave, lane, [&](int r) { return min(m0 + r, M - 1); }, [&](int r) { return min(n0 + r, N - 1); });
    }
    // piece I (0 .. PER-1) of chunk kc of this wavefront -> stage base `stage_lds` (bytes)
    template <int I>
    __device__ __forceinline__ void issue(int kc, unsigned stage_lds, int wave) const {
        if constexpr (I < CFG::A_PER) {
            glds16(a_off[I], a_base + (size_t)kc * (CFG::BK * 4), stage_lds + (unsigned)(wave * CFG::A_PER + I) * 1024u);
        } else {
            glds16(w_off[I - CFG::A_PER], w_base + (size_t)kc * (CFG::BK * 4),
                   stage_lds + CFG::W_BASE_BYTES + (unsigned)(wave * CFG::W_PER + (I - CFG::A_PER)) * 1024u);
        }
    }
};

template <class CFG, class SRC = LinearSource<CFG>>
struct GemmCore {
    struct Acc { f32x16 t[CFG::NT]; };
    struct Frag { f32x4 a[CFG::WM], b[CFG::WN]; };

    SRC src;
    unsigned lds0;                 // LDS byte address of the ring
    unsigned a_rd[2], w_rd[2];     // lane's fragment read addresses (stage 0, tile 0) for the two halves of a chunk
    int wave, lane;
    Frag F0, F1;

    __device__ __forceinline__ GemmCore(float* lds) {
        lds0 = SCP_LDS_ADDR(lds);
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int half = lane >> 5, l31 = lane & 31;
        const int ra = row_base() + l31, rw = col_base() + l31;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            a_rd[c] = lds0 + ra * 64 + 16 * ((2 * half + c) ^ ((ra >> 2) & 3));
            w_rd[c] = lds0 + CFG::W_BASE_BYTES + rw * 64 + 16 * ((2 * half + c) ^ ((rw >> 2) & 3));
        }
    }
    __device__ __forceinline__ int row_base() const { return 32 * CFG::WM * (wave / CFG::NWN); }
    __device__ __forceinline__ int col_base() const { return 32 * CFG::WN * (wave % CFG::NWN); }

    __device__ __forceinline__ void set_linear_sources(const float* A, const float* W, int m0, int n0, int M, int N, int K) {
        src.set(A, W, m0, n0, M, N, K, wave, lane);
    }

    // the interface csrc/gemm_core_split.h shares (vit_gemm.hip picks a core per launch)
    template <class FA, class FW>
    __device__ __forceinline__ void set_rows(const float* A, const void* W, int N, int K, FA a_row, FW w_row) {
        src.set_rows(A, static_cast<const float*>(W), K, wave, lane, a_row, w_row);
    }

    // fragment read R (0 .. NREAD-1) of half C of the chunk in stage S
    template <int S, int C, int R>
    __device__ __forceinline__ void read(Frag& f) {
        if constexpr (R < CFG::WM) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[R]) : "v"(a_rd[C]), "i"(S * CFG::STAGE_BYTES + R * 2048));
        } else {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.b[R - CFG::WM]) : "v"(w_rd[C]), "i"(S * CFG::STAGE_BYTES + (R - CFG::WM) * 2048));
        }
    }
    template <int N>
    __device__ __forceinline__ void mfma(Acc& acc, const Frag& f) {
        constexpr int j = N / CFG::NT, t = N % CFG::NT, ti = t / CFG::WN, tj = t % CFG::WN;
        // `s_nop 1`: hipcc does not pad hazards around asm statements, and it may materialise or copy an accumulator with
        // v_mov right in front of this statement (the zero fill lands in front of the first MFMA of a tile): a VALU write needs
        // two wait states before an MFMA reads the register as C (observed: stale first accumulator register on the
        // zero-trip path, K = 32).  Two issue cycles beside a 64-cycle MFMA are free.
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc.t[t]) : "v"(f.a[ti][j]), "v"(f.b[tj][j]));
    }

    static constexpr int imin(int a, int b) { return a < b ? a : b; }

    template <int S, int MODE>
    __device__ __forceinline__ void step(Acc& acc, int kc) {
        constexpr int NM = CFG::NM, NREAD = CFG::NREAD, PER = CFG::PER;
        constexpr int SN = (S + 1) % CFG::NSTAGE;
        // ---- first half: F0 holds its fragments; the second half's arrive in F1
        static_for<0, NM>([&](auto n) {
            constexpr int N = decltype(n)::value;
            mfma<N>(acc, F0);
            static_for<0, NREAD>([&](auto r) {
                constexpr int R = decltype(r)::value;
                if constexpr (imin(1 + 2 * R, NM - 1) == N) read<S, 1, R>(F1);
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)");
        // ---- second half
        static_for<0, NM>([&](auto n) {
            constexpr int N = decltype(n)::value;
            mfma<N>(acc, F1);
            if constexpr (MODE != GEMM_STEP_LAST) {
                if constexpr (N == 1) {
                    if constexpr (MODE == GEMM_STEP_ISSUE) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(PER * (CFG::NSTAGE - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
                static_for<0, NREAD>([&](auto r) {
                    constexpr int R = decltype(r)::value;
                    if constexpr (imin(2 + 2 * R, NM - 1) == N) read<SN, 0, R>(F0);
                });
                if constexpr (MODE == GEMM_STEP_ISSUE) {
                    static_for<0, PER>([&](auto p) {
                        constexpr int P = decltype(p)::value;
                        if constexpr (imin(3 + 4 * P, NM - 1) == N) src.template issue<P>(kc + CFG::NSTAGE, lds0 + S * CFG::STAGE_BYTES, wave);
                    });
                }
            }
        });
        if constexpr (MODE != GEMM_STEP_LAST) asm volatile("s_waitcnt lgkmcnt(0)");
    }

    // acc = sum over nk chunks; nk must be a positive multiple of NSTAGE (every K of the ViT and of the encoder's 3x3
    // layers is).  The control flow is a straight line -- prologue, one loop of NSTAGE-chunk trips, NSTAGE closing chunks --
    // so that the register allocator sees one live range per fragment and accumulator.  On return every LDS access and DMA
    // of this wavefront has completed and the accumulators are readable by ordinary code.
    __device__ __forceinline__ void run(Acc& acc, int nk) {
        static_for<0, CFG::NT>([&](auto t) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc.t[decltype(t)::value][r] = 0.f;
            asm volatile("" : "+v"(acc.t[decltype(t)::value]));      // the zero fill is materialised here, not in front of the first MFMA
        });
        // prologue
        static_for<0, CFG::NSTAGE>([&](auto s) {
            constexpr int S = decltype(s)::value;
            static_for<0, CFG::PER>([&](auto p) { src.template issue<decltype(p)::value>(S, lds0 + S * CFG::STAGE_BYTES, wave); });
        });
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(CFG::PER * (CFG::NSTAGE - 1)) : "memory");
        static_for<0, CFG::NREAD>([&](auto r) { read<0, 0, decltype(r)::value>(F0); });
        asm volatile("s_waitcnt lgkmcnt(0)");
        // steady state: NSTAGE chunks per trip so that the ring position is a compile-time constant
        for (int kc = 0; kc + CFG::NSTAGE < nk; kc += CFG::NSTAGE)
            static_for<0, CFG::NSTAGE>([&](auto s) { step<decltype(s)::value, GEMM_STEP_ISSUE>(acc, kc + decltype(s)::value); });
        // the last NSTAGE chunks: nothing left to fetch
        static_for<0, CFG::NSTAGE>([&](auto s) {
            constexpr int S = decltype(s)::value;
            step<S, S + 1 < CFG::NSTAGE ? GEMM_STEP_TAIL : GEMM_STEP_LAST>(acc, nk - CFG::NSTAGE + S);
        });
        // MFMA result -> VALU read: 16-pass instruction, 18 wait states (not padded by hipcc for asm statements)
        asm volatile("s_nop 15\n\ts_nop 3");
    }
};

}  // namespace scp
