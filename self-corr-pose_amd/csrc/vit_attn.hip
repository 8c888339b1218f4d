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

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;        // head dim (fixed: ViT-S)
constexpr int KT = 32;        // keys per tile
constexpr int KSTRIDE = 68;   // floats per K row in LDS (64 + 4 pad: 16-B aligned, bank-conflict free for b128)

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vit_attention_kernel(const float* __restrict__ qkv,
                                                                   float* __restrict__ out, int N, int H,
                                                                   float scale_log2e) {
    // K/V are staged SUB x 32 keys at a time (one barrier pair per SUB*64 MFMAs of every wavefront)
    constexpr int SUB = 2;
    __shared__ __attribute__((aligned(16))) float k_lds[SUB * KT * KSTRIDE];
    __shared__ __attribute__((aligned(16))) float v_lds[SUB * KT * HD];
    constexpr int THREADS = WAVES * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int q0 = (blockIdx.x * WAVES + wave) * 32;
    const size_t row_stride = (size_t)3 * H * HD;                 // floats between consecutive tokens
    const float* base = qkv + (size_t)b * N * row_stride + (size_t)h * HD;
    const float* kbase = base + (size_t)H * HD;
    const float* vbase = base + (size_t)2 * H * HD;

    // Q fragment: qreg[s] = Q[q][s + 32*half] * scale*log2(e)
    float qreg[32];
    {
        const int q = min(q0 + l31, N - 1);
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

    const int nstage = (N + SUB * KT - 1) / (SUB * KT);
    for (int st = 0; st < nstage; st++) {
        __syncthreads();
        for (int i = tid; i < SUB * KT * (HD / 4); i += THREADS) {
            const int r = i >> 4, c4 = i & 15;
            const int key = min(st * SUB * KT + r, N - 1);
            const float4 kv = reinterpret_cast<const float4*>(kbase + (size_t)key * row_stride)[c4];
            const float4 vv = reinterpret_cast<const float4*>(vbase + (size_t)key * row_stride)[c4];
            *reinterpret_cast<float4*>(k_lds + r * KSTRIDE + 4 * c4) = kv;
            *reinterpret_cast<float4*>(v_lds + r * HD + 4 * c4) = vv;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < SUB; sub++) {
            const int key_base = (st * SUB + sub) * KT;
            if (key_base >= N) break;   // uniform
            // ---- S^T = K Q^T : 32 k-steps of (d, d+32)
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; r++) s[r] = 0.f;
            {
                const float4* kp = reinterpret_cast<const float4*>(k_lds + (sub * KT + l31) * KSTRIDE + 32 * half);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float4 kk = kp[i];
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[4 * i + 0], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[4 * i + 1], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[4 * i + 2], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[4 * i + 3], s, 0, 0, 0);
                }
            }
            // ---- online softmax (lane owns 16 keys of its query; partner lane^32 owns the other 16)
            float m_tile = -INFINITY;
            if (key_base + KT > N) {   // ragged last tile: wavefront-uniform branch
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (key_base + acc_row(r, half) >= N) s[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) m_tile = fmaxf(m_tile, s[r]);
            m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
            const float m_new = fmaxf(m_run, m_tile);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; r++) { o_lo[r] *= alpha; o_hi[r] *= alpha; }
            // ---- O^T += V^T P^T : k-step r pairs key acc_row(r,0) with acc_row(r,1)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float* vrow = v_lds + (sub * KT + acc_row(r, half)) * HD + l31;
                o_lo = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[r], o_lo, 0, 0, 0);
                o_hi = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[r], o_hi, 0, 0, 0);
            }
        }
    }

    // ---- normalise and store: lane holds O[q = l31][d = acc_row(r, half) (+32)]
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    const int q = q0 + l31;
    if (q < N) {
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

extern "C" int scp_vit_attention_forward(const float* qkv, float* out, int B, int N, int H, int head_dim,
                                         float scale, void* stream) {
    if (B <= 0 || N <= 0 || H <= 0) return scp::fail(hipErrorInvalidValue, "vit_attention: empty problem");
    if (head_dim != HD) return scp::fail(hipErrorInvalidValue, "vit_attention: head_dim must be 64");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    const int qtiles = (N + 31) / 32;
    // 3 wavefronts per workgroup when that leaves no idle wavefront (1025 tokens = 33 tiles = 11 x 3)
    if (qtiles % 3 == 0 && qtiles % 4 != 0) {
        hipLaunchKernelGGL(vit_attention_kernel<3>, dim3(qtiles / 3, B * H), dim3(192), 0, st, qkv, out, N, H, sl);
    } else {
        hipLaunchKernelGGL(vit_attention_kernel<4>, dim3((qtiles + 3) / 4, B * H), dim3(256), 0, st, qkv, out, N, H, sl);
    }
    return scp::check_launch("vit_attention");
}
