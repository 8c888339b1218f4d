// self-corr-pose_amd/csrc/vit_attn.hip -- fused multi-head self-attention forward for the frozen DINO
// ViT-S/8 (N = 1025 tokens, 6 heads x 64), fp32 in / fp32 out, on the gfx950 fp32 matrix cores.
//
// Replaces Attention.forward of third-party/zsp/zsp/method/vision_transformer_flexible.py:85-101
// (q k^T * scale -> softmax -> @ v -> transpose/reshape), which materialises a [B,6,1025,1025] score
// tensor per block.  Flash-style: scores never leave registers.
//
// CDNA4 mapping (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles, A and B are ONE VGPR per lane):
//   * one wavefront owns 32 queries; per 32-key tile it computes the TRANSPOSED score tile
//     S^T[key][query] = K Q^T, so that lane (query = lane&31, half = lane>>5) holds 16 of the 32 keys
//     of ITS query in its accumulator registers: the online-softmax max/sum are lane-local, one
//     cross-lane exchange (lane ^ 32) per tile;
//   * the exponentiated accumulator registers are used AS IS as the B operand of the P.V MFMAs
//     (register r of both halves = one k-step of 2 keys), the matching A operand is the V row of those
//     keys read from LDS -- P is never moved, converted or written anywhere;
//   * the reduction index of Q K^T is paired as (d, d+32) so each lane reads 32 CONSECUTIVE floats of
//     its key row from LDS with 8 ds_read_b128 (row stride 68 floats: conflict-free);
//   * K/V tiles (8 KB each) are staged through LDS once per workgroup and shared by its wavefronts;
//   * exp2 with log2(e)*scale folded into Q (one v_exp_f32 per score).
// Roofline: 4*N^2*64 flop per (image, head) = 51.6 GFLOP per ViT block at B=32 -> 0.33 ms at the
// 155 TFLOP/s fp32 MFMA peak.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;        // head dim (fixed: ViT-S)
constexpr int KT = 32;        // keys per tile
constexpr float RESCALE_THR = 16.f;   // log2 units

// value of the partner lane (lane ^ 32) through v_permlane32_swap (gfx950): a register move between the two halves of
// the wavefront, instead of ds_bpermute's trip through the LDS crossbar and the lgkmcnt wait behind it
__device__ __forceinline__ float other_half(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    // r[0] = [x.lo | x.lo], r[1] = [x.hi | x.hi]: the partner's value is whichever differs from "mine"
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

#define SCP_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SCP_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// K/V tiles arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wavefront instruction, no staging
// VGPRs), double buffered: tile t+1 is in flight while tile t is consumed, one barrier per tile.
// The DMA destination is lane-linear, so the K image is made bank-conflict free by permuting the
// SOURCE address: 16-byte chunk c of key row r is stored in slot c ^ (r & 15) (and read back through
// the same XOR); V rows are read along d and need no swizzle.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 3) void vit_attention_kernel(const float* __restrict__ qkv,
                                                                   float* __restrict__ out, int N, int H,
                                                                   float scale_log2e, const int* __restrict__ q_rows,
                                                                   const int* __restrict__ q_count) {
    __shared__ __attribute__((aligned(16))) float k_lds[2][KT * HD];
    __shared__ __attribute__((aligned(16))) float v_lds[2][KT * HD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware placement: consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Put
    // all query groups of one (image, head) on ONE XCD so that its K/V (525 KB) is fetched into a single L2 instead
    // of up to eight (rocprofv3 FETCH_SIZE: 847 MB per launch before, against 151 MB of qkv).
    int bh = blockIdx.y, qg = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int slot = lin >> 3;
        bh = (slot / gridDim.x) * 8 + (lin & 7);
        qg = slot % gridDim.x;
    }
    const int b = bh / H, h = bh - b * H;
    const int q0 = (qg * WAVES + wave) * 32;
    // optional query selection (scp_vit_attention_forward_rows): query slot j of image b is token q_rows[b*N + j], only the
    // first q_count[b] slots exist.  Keys / values are always all N tokens.  A workgroup whose slots are all past the count
    // leaves before it touches anything; its outputs are simply not produced.
    const int n_query = q_count ? min(q_count[b], N) : N;
    if (qg * WAVES * 32 >= n_query) return;          // workgroup-uniform
    const size_t row_stride = (size_t)3 * H * HD;                 // floats between consecutive tokens
    const float* base = qkv + (size_t)b * N * row_stride + (size_t)h * HD;

    // LDS-DMA pieces: a K or V tile is 8 wavefront-instructions of 1 KiB (piece j: rows 4*(j&7) .. +3, 16 lanes x 16 B per
    // row); the 16 pieces of a (K tile, V tile) pair are dealt round-robin to the wavefronts.  Everything per-lane about a
    // piece is loop invariant -- row, swizzled chunk, K/V column block -- and is folded into ONE 32-bit element offset here;
    // per tile the source address is (uniform tile base) + that offset, so the loop carries no per-lane address arithmetic.
    constexpr int PIECES = (16 + WAVES - 1) / WAVES;
    const int ntiles = (N + KT - 1) / KT;
    const int r4 = lane >> 4, slot = lane & 15;
    unsigned lane_off[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; i++) {
        const int j = wave + WAVES * i, jj = j & 7, r = 4 * jj + r4;
        const bool is_v = j >= 8;
        const int chunk = is_v ? slot : (slot ^ (r & 15));
        lane_off[i] = (unsigned)r * (unsigned)row_stride + 4u * chunk + (is_v ? 2u : 1u) * (unsigned)(H * HD);
    }
    auto issue_piece = [&](int i, int kt, float* lds_tile) {
        const int j = wave + WAVES * i;
        const float* tile = base + (size_t)kt * KT * row_stride;                       // wavefront-uniform
        unsigned off = lane_off[i];
        if (kt == ntiles - 1) {   // ragged last tile (wavefront-uniform branch): rows past the sequence re-read the last key
            const int jj = j & 7, r = min(4 * jj + r4, N - 1 - kt * KT);
            const bool is_v = j >= 8;
            const int chunk = is_v ? slot : (slot ^ ((4 * jj + r4) & 15));
            off = (unsigned)r * (unsigned)row_stride + 4u * chunk + (is_v ? 2u : 1u) * (unsigned)(H * HD);
        }
        __builtin_amdgcn_global_load_lds(SCP_GLOBAL_PTR(tile + off), SCP_LDS_PTR(lds_tile + (j & 7) * 256), 16, 0, 0);
    };
    // K tile kt_k -> k_lds[bufk] and / or V tile kt_v -> v_lds[bufv]
    auto issue_tiles = [&](bool do_k, int kt_k, int bufk, bool do_v, int kt_v, int bufv) {
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            const int j = wave + WAVES * i;                                            // wavefront-uniform
            if (j >= 16) continue;
            if (j < 8) {
                if (do_k) issue_piece(i, kt_k, k_lds[bufk]);
            } else if (do_v) {
                issue_piece(i, kt_v, v_lds[bufv]);
            }
        }
    };

    // Q fragment: qreg[s] = Q[q][s + 32*half] * scale*log2(e)
    float qreg[32];
    {
        int q = min(q0 + l31, n_query - 1);
        if (q_rows) q = q_rows[(size_t)b * N + q];
        const float4* qp = reinterpret_cast<const float4*>(base + (size_t)q * row_stride + 32 * half);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 t = qp[i];
            qreg[4 * i + 0] = t.x * scale_log2e; qreg[4 * i + 1] = t.y * scale_log2e;
            qreg[4 * i + 2] = t.z * scale_log2e; qreg[4 * i + 3] = t.w * scale_log2e;
        }
    }
    f32x16 o_lo, o_hi;
#pragma unroll
    for (int r = 0; r < 16; r++) { o_lo[r] = 0.f; o_hi[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // Software pipeline inside the wavefront: the Q.K^T MFMAs of tile t+1 are issued BEFORE the
    // exponentials of tile t, so that those VALU/transcendental instructions execute in the shadow of the
    // matrix pipe (identical wavefronts on a SIMD run in lock-step, so cross-wavefront overlap alone
    // leaves the pipe idle during every softmax phase: measured 0.45 ms MFMA + 0.15 ms VALU, additive).
    // LDS rings: K(t+1) is read in iteration t from k_lds[(t+1)&1], V(t) from v_lds[t&1]; K(t+2) and
    // V(t+1) are in flight (LDS-DMA) meanwhile; one barrier per tile.
    const int kslot0 = (8 * half) ^ (l31 & 15);
    auto qk_tile = [&](int kbuf) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const float* krow = k_lds[kbuf] + l31 * HD;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 kk = *reinterpret_cast<const float4*>(krow + 4 * (kslot0 ^ i));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[4 * i + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[4 * i + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[4 * i + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[4 * i + 3], acc, 0, 0, 0);
        }
        return acc;
    };
    issue_tiles(true, 0, 0, true, 0, 0);
    if (ntiles > 1) issue_tiles(true, 1, 1, false, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 s = qk_tile(0);
    __syncthreads();   // K(0) has been read by every wavefront before iteration 0 refills k_lds[0] with K(2)
    auto tile_step = [&](int kt, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const int buf = kt & 1;
        // K(t+2) -> k_lds[t&1], V(t+1) -> v_lds[(t+1)&1]
        issue_tiles(kt + 2 < ntiles, kt + 2, buf, kt + 1 < ntiles, kt + 1, buf ^ 1);
        // V operands of tile t's P.V MFMAs: issued first so that their LDS latency is covered by the
        // max / rescale work below (the empty asm pins the loads here; the compiler otherwise sinks each
        // one to just before its MFMA and waits for it)
        float va[16], vb[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float* vrow = v_lds[buf] + acc_row(r, half) * HD + l31;
            va[r] = vrow[0];
            vb[r] = vrow[32];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) asm volatile("" : "+v"(va[r]), "+v"(vb[r]));
        // ---- running maximum of tile t (lane owns 16 keys of its query; partner lane^32 the other 16)
        const int key_base = kt * KT;
        float m_tile = -INFINITY;
        if (RAGGED) {   // only the peeled last tile carries the key-bound test
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (key_base + acc_row(r, half) >= N) s[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) m_tile = fmaxf(m_tile, s[r]);
        m_tile = fmaxf(m_tile, other_half(m_tile));
        // Deferred rescale: the running maximum is only raised (and O, l rescaled) when some query's tile
        // maximum exceeds it by more than 2^RESCALE_THR; otherwise the probabilities are taken against
        // the old maximum (bounded by 2^RESCALE_THR, harmless in fp32).  Wavefront-uniform decision; O, l
        // and the current P always share one scale, so the final O / l is the exact softmax.
        if (__any(m_tile > m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, m_tile);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; r++) { o_lo[r] *= alpha; o_hi[r] *= alpha; }
        }
        // ---- one straight-line region: Q.K^T of tile t+1 (matrix pipe) with the exponentials of tile t
        // in its shadow (VALU), then P.V of tile t.  (After the last tile the extra Q.K^T runs on a stale
        // K slot and is discarded -- keeping it unconditional keeps the region branch-free.)
        f32x16 s_next = qk_tile(buf ^ 1);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m_run);
            psum += s[r];
        }
        l_run += psum;
#pragma unroll
        for (int g = 0; g < 16; g++) {   // 2 MFMA then 3 VALU (sub, exp, add), 16 times
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        // ---- O^T += V^T P^T : k-step r pairs key acc_row(r,0) with acc_row(r,1)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            o_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], s[r], o_lo, 0, 0, 0);
            o_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[r], s[r], o_hi, 0, 0, 0);
        }
        s = s_next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's DMA pieces have landed
        __syncthreads();                                     // ... everybody's; ring slots may be reused
    };
    // the key-bound mask (15 compares + selects per lane) is only needed where a tile crosses the end of the sequence:
    // peel that tile so that the steady-state loop does not carry it
    const bool ragged = (N % KT) != 0;
    const int full = ragged ? ntiles - 1 : ntiles;
    for (int kt = 0; kt < full; kt++) tile_step(kt, std::false_type());
    if (ragged) tile_step(ntiles - 1, std::true_type());

    // ---- normalise and store: lane holds O[q = l31][d = acc_row(r, half) (+32)]
    const float l_tot = l_run + other_half(l_run);
    const float inv = 1.f / l_tot;
    const int qslot = q0 + l31;
    if (qslot < n_query) {
        const int q = q_rows ? q_rows[(size_t)b * N + qslot] : qslot;
        float* op = out + ((size_t)b * N + q) * (H * HD) + (size_t)h * HD;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int d = acc_row(r, half);
            op[d] = o_lo[r] * inv;
            op[d + 32] = o_hi[r] * inv;
        }
    }
}

}  // namespace

namespace {
int attention_impl(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale, const int* q_rows, const int* q_count,
                   void* stream) {
    if (B <= 0 || N <= 0 || H <= 0) return scp::fail(hipErrorInvalidValue, "vit_attention: empty problem");
    if (head_dim != HD) return scp::fail(hipErrorInvalidValue, "vit_attention: head_dim must be 64");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    const int qtiles = (N + 31) / 32;
    // 3 wavefronts per workgroup when that leaves no idle wavefront (1025 tokens = 33 tiles = 11 x 3)
    if (qtiles % 3 == 0 && qtiles % 4 != 0) {
        hipLaunchKernelGGL(vit_attention_kernel<3>, dim3(qtiles / 3, B * H), dim3(192), 0, st, qkv, out, N, H, sl, q_rows, q_count);
    } else {
        hipLaunchKernelGGL(vit_attention_kernel<4>, dim3((qtiles + 3) / 4, B * H), dim3(256), 0, st, qkv, out, N, H, sl, q_rows, q_count);
    }
    return scp::check_launch("vit_attention");
}
}  // namespace

extern "C" int scp_vit_attention_forward(const float* qkv, float* out, int B, int N, int H, int head_dim,
                                         float scale, void* stream) {
    return attention_impl(qkv, out, B, N, H, head_dim, scale, nullptr, nullptr, stream);
}

extern "C" int scp_vit_attention_forward_rows(const float* qkv, float* out, int B, int N, int H, int head_dim, float scale,
                                              const int* q_rows, const int* q_count, void* stream) {
    if (!q_rows || !q_count) return scp::fail(hipErrorInvalidValue, "vit_attention_rows: null selection");
    return attention_impl(qkv, out, B, N, H, head_dim, scale, q_rows, q_count, stream);
}
