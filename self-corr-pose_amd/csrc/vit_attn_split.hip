// self-corr-pose_amd/csrc/vit_attn_split.hip -- the ViT attention of csrc/vit_attn.hip (same operator, same fp32 accuracy) with
// both products on the bf16 matrix cores through EXACT operand splitting (csrc/gemm_core_split.h: every fp32 value = three bf16
// terms, the six leading partial products of nine accumulated in fp32, dropped terms < 2^-24 of |a b|).
//
// Replaces Attention.forward of third-party/zsp/zsp/method/vision_transformer_flexible.py:85-101 (q k^T * scale -> softmax -> @ v).
//
// Two launches per attention:
//   1. scp_vit_qkv_split: qkv [B,N,3,H,64] fp32 -> bf16 planes
//        Qp [3][B H][Npad][64]   (Q pre-multiplied by scale * log2(e) in fp32, then split),
//        Kp [3][B H][Npad][64],
//        Vt [3][B H][64][Npad]   TRANSPOSED (a row = one head dimension over the keys), keys permuted inside every block of 32 so
//                                that a lane's 8 keys of a P.V k-step are contiguous (see slot_key below);
//      Npad = N rounded up to 32, padding rows / columns are zeros.  One memory-bound pass (the next step is to have the qkv GEMM's
//      epilogue write these planes directly).
//   2. vit_attention_split_kernel: flash-style like csrc/vit_attn.hip -- one wavefront owns 32 queries, keeps the TRANSPOSED score
//      tile S^T[key][query] = K Q^T in its accumulator so that softmax statistics are lane-local, and feeds the exponentiated
//      accumulator straight back as the B operand of O^T += V^T P^T.  With v_mfma_f32_32x32x16_bf16 a lane's B operand is 8
//      consecutive k: accumulator registers 8 ks .. 8 ks + 7 of lane half hf are the keys slot_key(16 ks + 8 hf + i) of the tile --
//      the order Vt is stored in, so P needs no cross-lane movement: it is split in registers (3 x 4 VGPRs per k-step).
//      K / V planes arrive by LDS-DMA (double buffered, one barrier per tile), 16-B slots XOR-swizzled through the source address
//      so that every ds_read_b128 is bank-conflict free; Q's planes live in 48 VGPRs.
//      Per key tile and wavefront: 24 + 24 MFMAs of 32 cycles (fp32 cores: 32 + 32 of 64 cycles), 24 ds_read_b128,
//      ~90 VALU for the split of P besides the softmax itself.
// Roofline: 4 N^2 64 flop per (image, head) of fp32-equivalent work; six bf16 MFMA products per algorithmic one.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64, KT = 32;
constexpr float RESCALE_THR = 16.f;   // log2 units

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// key (0..31) stored at slot s of a 32-key block of Vt: slots 8 g .. 8 g + 7 are the accumulator rows of registers 8 (g >> 1) ..
// + 7 of lane half (g & 1)
__host__ __device__ inline int slot_key(int s) {
    const int g = s >> 3, i = s & 7;
    return 16 * (g >> 1) + 4 * (g & 1) + (i & 3) + 8 * (i >> 2);
}

__device__ __forceinline__ float other_half(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// x0, x1 -> packed (h, m, l) pairs with x = h + m + l exactly
struct Pair3 { unsigned h, m, l; };
__device__ __forceinline__ Pair3 split_pair(float x0, float x1) {
    Pair3 r;
    r.h = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(r.h << 16), r1 = x1 - __uint_as_float(r.h & 0xffff0000u);
    r.m = pack_bf16(r0, r1);
    r.l = pack_bf16(r0 - __uint_as_float(r.m << 16), r1 - __uint_as_float(r.m & 0xffff0000u));
    return r;
}

#define SCP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SCP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- 1. operand planes -------------------------------------------------------------------------------------------------------
// grid (Npad / 32, B H), 256 threads: thread = (token t = tid >> 3, 8 dims d0 = 8 (tid & 7)) for Q and K;
// for V the 32 x 64 tile goes through LDS and thread = (d = tid >> 2, slot group g = tid & 3) writes 8 keys of one dimension.
__global__ __launch_bounds__(256) void qkv_split_kernel(const float* __restrict__ qkv, __bf16* __restrict__ Qp, __bf16* __restrict__ Kp,
                                                        __bf16* __restrict__ Vt, int N, int Npad, int H, float scale_log2e, int v_only) {
    // v_only: Q / K planes were written by the qkv projection's epilogue (csrc/vit_gemm.hip scp_vit_linear_qkv); only V^T is made here
    __shared__ float vt[KT][HD + 1];
    const int tile = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int tid = threadIdx.x;
    const size_t row_stride = (size_t)3 * H * HD;
    const size_t plane_qk = (size_t)gridDim.y * Npad * HD, plane_v = plane_qk;
    {
        const int t = tid >> 3, d0 = 8 * (tid & 7);
        const int tok = tile * KT + t;
        float q[8], k[8], v[8];
        if (tok < N) {
            const float* src = qkv + ((size_t)b * N + tok) * row_stride + (size_t)h * HD + d0;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
                if (!v_only) {
                    a = *reinterpret_cast<const float4*>(src + 4 * i);
                    c = *reinterpret_cast<const float4*>(src + (size_t)H * HD + 4 * i);
                }
                const float4 e = *reinterpret_cast<const float4*>(src + (size_t)2 * H * HD + 4 * i);
                q[4 * i] = a.x * scale_log2e; q[4 * i + 1] = a.y * scale_log2e; q[4 * i + 2] = a.z * scale_log2e; q[4 * i + 3] = a.w * scale_log2e;
                k[4 * i] = c.x; k[4 * i + 1] = c.y; k[4 * i + 2] = c.z; k[4 * i + 3] = c.w;
                v[4 * i] = e.x; v[4 * i + 1] = e.y; v[4 * i + 2] = e.z; v[4 * i + 3] = e.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) q[i] = k[i] = v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) vt[t][d0 + i] = v[i];
        if (!v_only) {
            u32x4 qh, qm, ql, kh, km, kl;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const Pair3 a = split_pair(q[2 * p], q[2 * p + 1]), c = split_pair(k[2 * p], k[2 * p + 1]);
                qh[p] = a.h; qm[p] = a.m; ql[p] = a.l;
                kh[p] = c.h; km[p] = c.m; kl[p] = c.l;
            }
            const size_t o = ((size_t)bh * Npad + tok) * HD + d0;
            *reinterpret_cast<u32x4*>(Qp + o) = qh;
            *reinterpret_cast<u32x4*>(Qp + plane_qk + o) = qm;
            *reinterpret_cast<u32x4*>(Qp + 2 * plane_qk + o) = ql;
            *reinterpret_cast<u32x4*>(Kp + o) = kh;
            *reinterpret_cast<u32x4*>(Kp + plane_qk + o) = km;
            *reinterpret_cast<u32x4*>(Kp + 2 * plane_qk + o) = kl;
        }
    }
    __syncthreads();
    {
        const int d = tid >> 2, g = tid & 3;
        u32x4 vh, vm, vl;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const Pair3 a = split_pair(vt[slot_key(8 * g + 2 * p)][d], vt[slot_key(8 * g + 2 * p + 1)][d]);
            vh[p] = a.h; vm[p] = a.m; vl[p] = a.l;
        }
        const size_t o = ((size_t)bh * HD + d) * Npad + (size_t)tile * KT + 8 * g;
        *reinterpret_cast<u32x4*>(Vt + o) = vh;
        *reinterpret_cast<u32x4*>(Vt + plane_v + o) = vm;
        *reinterpret_cast<u32x4*>(Vt + 2 * plane_v + o) = vl;
    }
}

// ---- 2. attention ------------------------------------------------------------------------------------------------------------
// EXACT: six partial products on exactly split operands; else one product on operands rounded to bf16 (the h planes alone:
// BASELINE configs[4] precision, softmax statistics still fp32)
template <int WAVES, bool EXACT>
__global__ __launch_bounds__(WAVES * 64, 2) void vit_attention_split_kernel(const __bf16* __restrict__ Qp, const __bf16* __restrict__ Kp,
                                                                         const __bf16* __restrict__ Vt, float* __restrict__ out, int N,
                                                                         int Npad, int H, const int* __restrict__ q_rows,
                                                                         const int* __restrict__ q_count, int n_main,
                                                                         __bf16* __restrict__ out3) {
    scp::claim_vgprs<256>();                                    // bf16 MFMAs: two wavefronts fill a SIMD's register file (scp_common.h)
    // per buffer: K planes 3 x [32 keys][64 d] bf16 (128 B rows), V^T planes 3 x [64 d][32 keys] bf16 (64 B rows)
    __shared__ __attribute__((aligned(16))) char k_lds[2][3 * 4096];
    __shared__ __attribute__((aligned(16))) char v_lds[2][3 * 4096];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // all query groups of one (image, head) on ONE XCD (see csrc/vit_attn.hip)
    int bh = blockIdx.y, qg = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int slot = lin >> 3;
        bh = (slot / gridDim.x) * 8 + (lin & 7);
        qg = slot % gridDim.x;
    }
    const int b = bh / H, h = bh - b * H;
    const int q0 = (qg * WAVES + wave) * 32;
    const int n_query = q_count ? min(q_count[b], N) : n_main;    // n_main < N: the last few queries run in the tail kernel
    if (qg * WAVES * 32 >= n_query) return;          // workgroup-uniform
    const size_t plane = (size_t)gridDim.y * Npad * HD;      // elements per plane (gridDim.y = B H also in the remapped order)
    const __bf16* kbase = Kp + (size_t)bh * Npad * HD;
    const __bf16* vbase = Vt + (size_t)bh * HD * Npad;
    const int ntiles = Npad / KT;

    // LDS-DMA pieces (1 KiB each): K plane tile = 4 pieces of 8 keys x 128 B (lane: key = lane >> 3, slot = lane & 7),
    // V^T plane tile = 4 pieces of 16 dims x 64 B (lane: d = lane >> 2, slot = lane & 3); 12 + 12 pieces per (K, V) tile pair,
    // dealt round-robin to the wavefronts.  The 16-B slot a lane FETCHES is permuted so that the lane-linear destination holds
    // slot s of row r at s ^ f(r): f = (key >> 1) & 7 for K, (d >> 2) & 3 for V (conflict-free b128 reads below).
    constexpr int PIECES = (24 + WAVES - 1) / WAVES;
    unsigned lane_off[PIECES];      // element offset inside the (bh) matrix of the plane, tile 0
#pragma unroll
    for (int i = 0; i < PIECES; i++) {
        const int j = wave + WAVES * i;
        if (j < 12) {
            const int pl = j >> 2, key = 8 * (j & 3) + (lane >> 3), s = (lane & 7) ^ ((key >> 1) & 7);
            lane_off[i] = (unsigned)(pl * plane) + (unsigned)key * HD + 8u * s;
        } else {
            const int jj = j - 12, pl = jj >> 2, d = 16 * (jj & 3) + (lane >> 2), s = (lane & 3) ^ ((d >> 2) & 3);
            lane_off[i] = (unsigned)(pl * plane) + (unsigned)d * (unsigned)Npad + 8u * s;
        }
    }
    auto issue_tiles = [&](bool do_k, int kt_k, int bufk, bool do_v, int kt_v, int bufv) {
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            const int j = wave + WAVES * i;                                            // wavefront-uniform
            if (j >= 24) continue;
            if (j < 12) {
                if (do_k)
                    __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(kbase + (size_t)kt_k * KT * HD + lane_off[i]),
                                                     SCP_LDS_PTR(k_lds[bufk] + j * 1024), 16, 0, 0);
            } else if (do_v) {
                __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(vbase + (size_t)kt_v * KT + lane_off[i]),
                                                 SCP_LDS_PTR(v_lds[bufv] + (j - 12) * 1024), 16, 0, 0);
            }
        }
    };

    // Q planes: B operand of K Q^T, step s covers dims 16 s .. 16 s + 15, lane half hf the dims 16 s + 8 hf .. + 7
    bf16x8 qf[3][4];
    {
        int q = min(q0 + l31, n_query - 1);
        if (q_rows) q = q_rows[(size_t)b * N + q];
        const __bf16* qp = Qp + ((size_t)bh * Npad + q) * HD + 8 * half;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int s = 0; s < 4; s++) qf[p][s] = *reinterpret_cast<const bf16x8*>(qp + (size_t)p * plane + 16 * s);
    }
    f32x16 o_lo, o_hi;
#pragma unroll
    for (int r = 0; r < 16; r++) { o_lo[r] = 0.f; o_hi[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // fragment read offsets: K: key l31, slot 2 s + half; V^T: dim l31 (+ 32), slot 2 ks + half
    const int k_row = l31 * 128, k_sw = (l31 >> 1) & 7;
    const int v_row0 = l31 * 64, v_sw0 = (l31 >> 2) & 3;              // d = l31
    const int v_row1 = (l31 + 32) * 64, v_sw1 = ((l31 + 32) >> 2) & 3;  // d = l31 + 32
    auto mma = [](const bf16x8& a, const bf16x8& bq, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, c, 0, 0, 0); };
    // K fragments of steps sa, sa + 1 of the tile in k_lds[kbuf] (6 ds_read_b128); V^T fragments of k-step ks (6 ds_read_b128)
    struct KFrag { bf16x8 h[2], m[2], l[2]; };
    struct VFrag { bf16x8 h[2], m[2], l[2]; };       // [0]: d = l31, [1]: d = l31 + 32
    auto load_k = [&](int kbuf, int sa) {
        KFrag f;
        const char* kb = k_lds[kbuf] + k_row;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int off = 16 * ((2 * (sa + i) + half) ^ k_sw);
            f.h[i] = *reinterpret_cast<const bf16x8*>(kb + off);
            f.m[i] = *reinterpret_cast<const bf16x8*>(kb + 4096 + off);
            f.l[i] = *reinterpret_cast<const bf16x8*>(kb + 8192 + off);
        }
        return f;
    };
    auto load_v = [&](int vbuf, int ks) {
        VFrag f;
        const char* vb = v_lds[vbuf];
        const int off0 = v_row0 + 16 * ((2 * ks + half) ^ v_sw0), off1 = v_row1 + 16 * ((2 * ks + half) ^ v_sw1);
        f.h[0] = *reinterpret_cast<const bf16x8*>(vb + off0); f.m[0] = *reinterpret_cast<const bf16x8*>(vb + 4096 + off0);
        f.l[0] = *reinterpret_cast<const bf16x8*>(vb + 8192 + off0);
        f.h[1] = *reinterpret_cast<const bf16x8*>(vb + off1); f.m[1] = *reinterpret_cast<const bf16x8*>(vb + 4096 + off1);
        f.l[1] = *reinterpret_cast<const bf16x8*>(vb + 8192 + off1);
        return f;
    };
    // 12 MFMAs: steps sa, sa + 1 into two independent accumulator chains (smallest terms first)
    auto qk_half = [&](const KFrag& f, int sa, f32x16& acc0, f32x16& acc1) {
        if constexpr (EXACT) {
            acc0 = mma(f.m[0], qf[1][sa], acc0); acc1 = mma(f.m[1], qf[1][sa + 1], acc1);
            acc0 = mma(f.l[0], qf[0][sa], acc0); acc1 = mma(f.l[1], qf[0][sa + 1], acc1);
            acc0 = mma(f.h[0], qf[2][sa], acc0); acc1 = mma(f.h[1], qf[2][sa + 1], acc1);
            acc0 = mma(f.m[0], qf[0][sa], acc0); acc1 = mma(f.m[1], qf[0][sa + 1], acc1);
            acc0 = mma(f.h[0], qf[1][sa], acc0); acc1 = mma(f.h[1], qf[1][sa + 1], acc1);
        }
        acc0 = mma(f.h[0], qf[0][sa], acc0); acc1 = mma(f.h[1], qf[0][sa + 1], acc1);
    };
    auto pv_step = [&](const VFrag& f, const bf16x8& Ph, const bf16x8& Pm, const bf16x8& Pl) {
        if constexpr (EXACT) {
            o_lo = mma(f.m[0], Pm, o_lo); o_hi = mma(f.m[1], Pm, o_hi);
            o_lo = mma(f.l[0], Ph, o_lo); o_hi = mma(f.l[1], Ph, o_hi);
            o_lo = mma(f.h[0], Pl, o_lo); o_hi = mma(f.h[1], Pl, o_hi);
            o_lo = mma(f.m[0], Ph, o_lo); o_hi = mma(f.m[1], Ph, o_hi);
            o_lo = mma(f.h[0], Pm, o_lo); o_hi = mma(f.h[1], Pm, o_hi);
        }
        o_lo = mma(f.h[0], Ph, o_lo); o_hi = mma(f.h[1], Ph, o_hi);
    };
    auto qk_tile = [&](int kbuf) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = 0.f;
        qk_half(load_k(kbuf, 0), 0, acc0, acc1);
        qk_half(load_k(kbuf, 2), 2, acc0, acc1);
#pragma unroll
        for (int r = 0; r < 16; r++) acc0[r] += acc1[r];
        return acc0;
    };
    // scheduling fences: the compiler otherwise sinks every ds_read next to its MFMA and waits for it there (12 exposed LDS
    // latencies per tile with 1.5 wavefronts per SIMD to cover them); here each group of 6 reads is issued a whole MFMA group early
    auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
#define SCP_INTERLEAVE12(VALU, DS)                                        \
    do {                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);              \
        _Pragma("unroll") for (int g_ = 0; g_ < 12; g_++) {              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
            __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);        \
        }                                                                \
    } while (0)

    issue_tiles(true, 0, 0, true, 0, 0);
    if (ntiles > 1) issue_tiles(true, 1, 1, false, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s = qk_tile(0);
    __syncthreads();   // K(0) has been read by every wavefront before iteration 0 refills k_lds[0] with K(2)
    auto tile_step = [&](int kt, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const int buf = kt & 1;
        issue_tiles(kt + 2 < ntiles, kt + 2, buf, kt + 1 < ntiles, kt + 1, buf ^ 1);
        // ---- region 0: first half of K(t+1)'s fragments on their way while the running maximum is taken
        const KFrag ka = load_k(buf ^ 1, 0);
        const int key_base = kt * KT;
        float m_tile = -INFINITY;
        if (RAGGED) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (key_base + acc_row(r, half) >= N) s[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) m_tile = fmaxf(m_tile, s[r]);
        m_tile = fmaxf(m_tile, other_half(m_tile));
        if (__any(m_tile > m_run + RESCALE_THR)) {       // deferred rescale (csrc/vit_attn.hip)
            const float m_new = fmaxf(m_run, m_tile);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; r++) { o_lo[r] *= alpha; o_hi[r] *= alpha; }
        }
        fence();
        // ---- region 1: Q.K^T(t+1) steps 0,1 | reads of steps 2,3 | exponentials of tile t
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = 0.f;
        const KFrag kb2 = load_k(buf ^ 1, 2);
        qk_half(ka, 0, acc0, acc1);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m_run);
            psum += s[r];
        }
        l_run += psum;
        SCP_INTERLEAVE12(4, 6);
        fence();
        // ---- region 2: Q.K^T(t+1) steps 2,3 | reads of V(t) k-step 0 | split of P
        const VFrag va = load_v(buf, 0);
        qk_half(kb2, 2, acc0, acc1);
        u32x4 ph[2], pm[2], pl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const Pair3 a = split_pair(s[8 * ks + 2 * p], s[8 * ks + 2 * p + 1]);
                ph[ks][p] = a.h; pm[ks][p] = a.m; pl[ks][p] = a.l;
            }
        SCP_INTERLEAVE12(7, 6);
        fence();
        // ---- region 3: O^T += V^T P^T k-step 0 | reads of k-step 1 | the two Q.K^T chains summed
        const VFrag vb2 = load_v(buf, 1);
        pv_step(va, __builtin_bit_cast(bf16x8, ph[0]), __builtin_bit_cast(bf16x8, pm[0]), __builtin_bit_cast(bf16x8, pl[0]));
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = acc0[r] + acc1[r];
        SCP_INTERLEAVE12(1, 6);
        fence();
        // ---- region 4: k-step 1
        pv_step(vb2, __builtin_bit_cast(bf16x8, ph[1]), __builtin_bit_cast(bf16x8, pm[1]), __builtin_bit_cast(bf16x8, pl[1]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const bool ragged = (N % KT) != 0;
    const int full = ragged ? ntiles - 1 : ntiles;
    for (int kt = 0; kt < full; kt++) tile_step(kt, std::false_type());
    if (ragged) tile_step(ntiles - 1, std::true_type());

    const float l_tot = l_run + other_half(l_run);
    const float inv = 1.f / l_tot;
    const int qslot = q0 + l31;
    if (out && qslot < n_query) {
        const int q = q_rows ? q_rows[(size_t)b * N + qslot] : qslot;
        float* op = out + ((size_t)b * N + q) * (H * HD) + (size_t)h * HD;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int d = acc_row(r, half);
            op[d] = o_lo[r] * inv;
            op[d + 32] = o_hi[r] * inv;
        }
    }
    if (out3) {
        // the result as the proj GEMM's pre-split A operand (TILED planes [rows / 32][H 64 / 16][3][32][16] bf16 of the [B N][H 64]
        // matrix, csrc/gemm_core_split.h): the wavefront's O^T block goes through LDS (the K / V buffers are idle: every wavefront
        // left the tile loop through its closing barrier) and is read back query-row-wise -- 8 consecutive dims per lane, one exact
        // split, three 16-byte stores.
        constexpr int OS = 68;                                   // row stride in floats: 16-byte aligned rows, conflict-free b128 writes
        float* st = reinterpret_cast<float*>(wave < 2 ? k_lds[0] : v_lds[0]) + (wave & 1) * (32 * OS);
        static_assert(2 * 32 * OS * 4 <= 2 * 3 * 4096, "two staging blocks per buffer pair");
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            // registers 4 g4 .. 4 g4 + 3 = dims 8 g4 + 4 half + (0..3) of the lane's query
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const f32x4_t lo = {o_lo[4 * g4] * inv, o_lo[4 * g4 + 1] * inv, o_lo[4 * g4 + 2] * inv, o_lo[4 * g4 + 3] * inv};
            const f32x4_t hi = {o_hi[4 * g4] * inv, o_hi[4 * g4 + 1] * inv, o_hi[4 * g4 + 2] * inv, o_hi[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4_t*>(st + l31 * OS + 8 * g4 + 4 * half) = lo;
            *reinterpret_cast<f32x4_t*>(st + l31 * OS + 32 + 8 * g4 + 4 * half) = hi;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int ql = lane >> 1, qs = q0 + ql;
        if (qs < n_query) {
            const int q = q_rows ? q_rows[(size_t)b * N + qs] : qs;
            const int row = b * N + q, kch = (H * HD) >> 4;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int d8 = 16 * c + 8 * (lane & 1);
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t a = *reinterpret_cast<const f32x4_t*>(st + ql * OS + d8), e = *reinterpret_cast<const f32x4_t*>(st + ql * OS + d8 + 4);
                u32x4 ph, pm, pl;
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const Pair3 s0 = split_pair(a[2 * p], a[2 * p + 1]), s1 = split_pair(e[2 * p], e[2 * p + 1]);
                    ph[p] = s0.h; pm[p] = s0.m; pl[p] = s0.l;
                    ph[2 + p] = s1.h; pm[2 + p] = s1.m; pl[2 + p] = s1.l;
                }
                const int k = h * HD + d8;
                __bf16* dst = out3 + ((((size_t)(row >> 5) * kch + (k >> 4)) * 3) << 9) + ((row & 31) << 4) + (k & 15);
                *reinterpret_cast<u32x4*>(dst) = ph;
                *reinterpret_cast<u32x4*>(dst + 512) = pm;
                *reinterpret_cast<u32x4*>(dst + 1024) = pl;
            }
        }
    }
}


// ---- 3. the few queries beyond the last full tile of 32 ---------------------------------------------------------------------------
// N = 1025 = 32 x 32 + 1: a 33rd query tile holds ONE query (the class token) per (image, head), and a workgroup that streams all
// keys and values for it costs as much as one that serves 128 queries -- a ninth of the launch.  Those r = N % 32 <= 8 queries
// run here instead, in plain fp32 FMA arithmetic on the qkv tensor itself (25 MFLOP in all), flash-decoding style so that the
// launch fills the machine: TAIL_CHUNKS workgroups per (query, image, head) each take a range of <= 256 keys -- thread = key for
// the scores and the chunk's softmax statistics, thread = (4 dims, 1 of 16 key parts) for sum_j p_j v_j -- and leave
// (max, sum, partial output[64]); a second tiny kernel merges the chunks.
constexpr int TAIL_CHUNKS = 8, TAIL_REC = 2 + HD;

// Qp != nullptr: Q (pre-scaled) comes from the operand planes (x = h + m + l exactly: the sum reproduces the fp32 value q * scale the
// other branch computes, so both give the same bits); K and V always from the qkv tensor
__device__ __forceinline__ float4 planes4(const __bf16* p, size_t plane) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(p), m = *reinterpret_cast<const bf16x4*>(p + plane),
                 l = *reinterpret_cast<const bf16x4*>(p + 2 * plane);
    return make_float4(((float)h[0] + (float)m[0]) + (float)l[0], ((float)h[1] + (float)m[1]) + (float)l[1],
                       ((float)h[2] + (float)m[2]) + (float)l[2], ((float)h[3] + (float)m[3]) + (float)l[3]);
}

__global__ __launch_bounds__(256) void attention_tail_partial_kernel(const float* __restrict__ qkv, float* __restrict__ partial, int N, int H,
                                                                     float scale_log2e, int first_query, int chunk_keys,
                                                                     const __bf16* __restrict__ Qp, const __bf16* __restrict__ Kp, size_t plane,
                                                                     int Npad) {
    __shared__ __attribute__((aligned(16))) float p_lds[256];
    __shared__ __attribute__((aligned(16))) float o_lds[16 * HD];
    __shared__ float red[8];
    const int c = blockIdx.x, qi = blockIdx.y, bh = blockIdx.z, b = bh / H, h = bh - b * H;
    const int q = first_query + qi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row_stride = (size_t)3 * H * HD;
    const float* base = qkv + (size_t)b * N * row_stride + (size_t)h * HD;
    const int k0 = c * chunk_keys, nk = max(0, min(chunk_keys, N - k0));
    // scores of the chunk's keys: a key row (64 floats = 256 B) is read by the 16 lanes of a DPP row, one float4 each -- every load
    // instruction covers four whole rows (a thread reading its own row touched 64 cache lines per instruction) -- and the 16 partial
    // dot products are summed inside the row; lane d4 of row-group `part` keeps key part + 16 d4.
    const int d4q = tid & 15, partq = tid >> 4, kidx = partq + 16 * d4q;
    float s = -INFINITY;
    {
        float4 u;
        if (Qp) u = planes4(Qp + ((size_t)bh * Npad + q) * HD + 4 * d4q, plane);        // pre-scaled query row from its planes
        else {
            u = reinterpret_cast<const float4*>(base + (size_t)q * row_stride)[d4q];
            u.x *= scale_log2e; u.y *= scale_log2e; u.z *= scale_log2e; u.w *= scale_log2e;
        }
        // keys from the fp32 K third that the qkv projection keeps for this kernel
        const float* kbase = base + (size_t)H * HD + 4 * d4q;
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = partq + 16 * jj;
            float a = 0.f;
            if (j < nk) {
                const float4 t = *reinterpret_cast<const float4*>(kbase + (size_t)(k0 + j) * row_stride);
                a = fmaf(u.x, t.x, a); a = fmaf(u.y, t.y, a);
                a = fmaf(u.z, t.z, a); a = fmaf(u.w, t.w, a);
            }
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, true));     // lane ^ 1
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xF, 0xF, true));     // lane ^ 2
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x141, 0xF, 0xF, true));    // row_half_mirror
            a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x140, 0xF, 0xF, true));    // row_mirror
            if (d4q == jj && j < nk) s = a;
        }
    }
    float m = s;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, __shfl_xor(m, k));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float e = kidx < nk ? __builtin_amdgcn_exp2f(s - m) : 0.f;
    p_lds[kidx] = e;
    float l = e;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) l += __shfl_xor(l, k);
    if (lane == 0) red[4 + wave] = l;
    __syncthreads();
    l = (red[4] + red[5]) + (red[6] + red[7]);
    const int d4 = tid & 15, part = tid >> 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* vbase = base + (size_t)2 * H * HD + 4 * d4;
    for (int j = part; j < nk; j += 16) {
        const float pj = p_lds[j];
        const float4 v = *reinterpret_cast<const float4*>(vbase + (size_t)(k0 + j) * row_stride);
        acc.x = fmaf(pj, v.x, acc.x); acc.y = fmaf(pj, v.y, acc.y); acc.z = fmaf(pj, v.z, acc.z); acc.w = fmaf(pj, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(o_lds + part * HD + 4 * d4) = acc;
    __syncthreads();
    float* rec = partial + (((size_t)bh * gridDim.y + qi) * TAIL_CHUNKS + c) * TAIL_REC;
    if (tid < HD) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += o_lds[k * HD + tid];
        rec[2 + tid] = t;
    }
    if (tid == 0) {
        rec[0] = nk > 0 ? m : -INFINITY;
        rec[1] = nk > 0 ? l : 0.f;
    }
}

__global__ __launch_bounds__(HD) void attention_tail_merge_kernel(const float* __restrict__ partial, float* __restrict__ out, int N, int H,
                                                                  int first_query, __bf16* __restrict__ out3) {
    const int qi = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H, d = threadIdx.x;
    const float* rec = partial + ((size_t)bh * gridDim.x + qi) * TAIL_CHUNKS * TAIL_REC;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < TAIL_CHUNKS; c++) m = fmaxf(m, rec[c * TAIL_REC]);
    float l = 0.f, o = 0.f;
#pragma unroll
    for (int c = 0; c < TAIL_CHUNKS; c++) {
        const float w = __builtin_amdgcn_exp2f(rec[c * TAIL_REC] - m);      // exp2(-inf) = 0 for an empty chunk
        l = fmaf(rec[c * TAIL_REC + 1], w, l);
        o = fmaf(rec[c * TAIL_REC + 2 + d], w, o);
    }
    const float y = o / l;
    if (out) out[((size_t)b * N + first_query + qi) * (H * HD) + (size_t)h * HD + d] = y;
    if (out3) {                                                   // the same element of the tiled planes (see the main kernel)
        const int row = b * N + first_query + qi, k = h * HD + d, kch = (H * HD) >> 4;
        const __bf16 yh = (__bf16)y;
        const float r1 = y - (float)yh;
        const __bf16 ym = (__bf16)r1;
        __bf16* dst = out3 + ((((size_t)(row >> 5) * kch + (k >> 4)) * 3) << 9) + ((row & 31) << 4) + (k & 15);
        dst[0] = yh; dst[512] = ym; dst[1024] = (__bf16)(r1 - (float)ym);
    }
}

}  // namespace

extern "C" size_t scp_vit_attention_split_workspace(int B, int N, int H) {
    const size_t npad = (size_t)((N + KT - 1) / KT) * KT;
    // operand planes + the tail queries' chunk records (<= 8 queries x TAIL_CHUNKS per (image, head))
    return (size_t)9 * B * H * npad * HD * sizeof(__bf16) + (size_t)B * H * 8 * TAIL_CHUNKS * TAIL_REC * sizeof(float);
}

namespace {
int attention_split_impl(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale, const int* q_rows, const int* q_count,
                         int exact, int presplit_qk, void* workspace, size_t workspace_bytes, void* stream, void* out_planes = nullptr);
}
extern "C" int scp_vit_attention_split_forward(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale,
                                               const int* q_rows, const int* q_count, int exact, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    return attention_split_impl(qkv, out, B, N, H, head_dim, scale, q_rows, q_count, exact, 0, workspace, workspace_bytes, stream);
}
extern "C" int scp_vit_attention_split_forward_presplit(const float* qkv, float* out, void* out_planes, int B, int N, int H, int head_dim,
                                                        float scale, const int* q_rows, const int* q_count, void* workspace,
                                                        size_t workspace_bytes, void* stream) {
    return attention_split_impl(qkv, out, B, N, H, head_dim, scale, q_rows, q_count, 1, 1, workspace, workspace_bytes, stream, out_planes);
}
namespace {
int attention_split_impl(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale, const int* q_rows, const int* q_count,
                         int exact, int presplit_qk, void* workspace, size_t workspace_bytes, void* stream, void* out_planes) {
    if (B <= 0 || N <= 0 || H <= 0) return scp::fail(hipErrorInvalidValue, "vit_attention_split: empty problem");
    if (head_dim != HD) return scp::fail(hipErrorInvalidValue, "vit_attention_split: head_dim must be 64");
    if (!qkv || (!out && !out_planes) || !workspace || (q_rows == nullptr) != (q_count == nullptr))
        return scp::fail(hipErrorInvalidValue, "vit_attention_split: null argument");
    if (out_planes && (!exact || (size_t)3 * (((size_t)B * N + 31) / 32 * 32) * H * HD >= (1ull << 31)))
        return scp::fail(hipErrorInvalidValue, "vit_attention_split: output planes need the exact split and < 2^31 elements");
    __bf16* out3 = static_cast<__bf16*>(out_planes);
    if (workspace_bytes < scp_vit_attention_split_workspace(B, N, H))
        return scp::fail(hipErrorInvalidValue, "vit_attention_split: workspace too small");
    const int npad = ((N + KT - 1) / KT) * KT;
    const size_t plane3 = (size_t)3 * B * H * npad * HD;
    if (plane3 >= (1ull << 31)) return scp::fail(hipErrorInvalidValue, "vit_attention_split: operand planes larger than 2^31 elements");
    __bf16* Qp = static_cast<__bf16*>(workspace);
    __bf16* Kp = Qp + plane3;
    __bf16* Vt = Kp + plane3;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(qkv_split_kernel, dim3(npad / KT, B * H), dim3(256), 0, st, qkv, Qp, Kp, Vt, N, npad, H, sl, presplit_qk);
    // four wavefronts per workgroup, two workgroups per CU = two wavefronts per SIMD (three per workgroup fit 1025 tokens without
    // an idle wavefront, but leave every other SIMD with a single wavefront and nothing to cover its stalls: 400 vs 368 us)
    int qtiles = (N + 31) / 32;
    const int tail = N % 32;
    // the queries beyond the last full tile, when they are few, go to attention_tail_queries_kernel (see there): for N = 1025 the
    // main launch is then exactly 3.0 rounds of 512 workgroups instead of 3.375
    const int chunk_keys = (N + TAIL_CHUNKS - 1) / TAIL_CHUNKS;
    const bool split_tail = !q_rows && exact && tail >= 1 && tail <= 8 && N >= 64 && chunk_keys <= 256;
    if (split_tail) qtiles -= 1;
    const dim3 grid((qtiles + 3) / 4, B * H);
    // n_query of the main kernel: with the tail split off only the full tiles' queries
    const int n_main = split_tail ? N - tail : N;
    if (exact) hipLaunchKernelGGL((vit_attention_split_kernel<4, true>), grid, dim3(256), 0, st, Qp, Kp, Vt, out, N, npad, H, q_rows, q_count, n_main, out3);
    else hipLaunchKernelGGL((vit_attention_split_kernel<4, false>), grid, dim3(256), 0, st, Qp, Kp, Vt, out, N, npad, H, q_rows, q_count, n_main, out3);
    if (split_tail) {
        float* partial = reinterpret_cast<float*>(Vt + plane3);
        hipLaunchKernelGGL(attention_tail_partial_kernel, dim3(TAIL_CHUNKS, tail, B * H), dim3(256), 0, st, qkv, partial, N, H, sl, N - tail,
                           chunk_keys, presplit_qk ? Qp : nullptr, Kp, plane3 / 3, npad);
        hipLaunchKernelGGL(attention_tail_merge_kernel, dim3(tail, B * H), dim3(HD), 0, st, partial, out, N, H, N - tail, out3);
    }
    return scp::check_launch("vit_attention_split");
}
}  // namespace
