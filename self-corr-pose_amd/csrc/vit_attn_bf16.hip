// self-corr-pose_amd/csrc/vit_attn_bf16.hip -- the ViT attention for BASELINE configs[4] precision (mixed bf16): bf16 q/k/v in,
// bf16 out, bf16 matrix cores with fp32 accumulation, softmax statistics in fp32.
//
// Same operator as csrc/vit_attn.hip (Attention.forward of vision_transformer_flexible.py:85-101) and the same mapping --
// one wavefront per 32 queries, transposed score tile S^T[key][query] = K Q^T so that the online-softmax statistics are
// lane-local and the exponentiated accumulators feed the P.V product without leaving their registers -- on
// v_mfma_f32_32x32x16_bf16 (8 bf16 per lane per operand):
//   * Q.K^T: 4 MFMAs per 32-key tile (k = head dim in steps of 16); K rows arrive by LDS-DMA, 16-byte chunks XOR-swizzled
//     by (row >> 1) & 7 through the SOURCE address so that the ds_read_b128 of 16 consecutive rows hit 16 different banks;
//   * P.V: the k-slot (half, j) of step t is key (j&3) + 16t + 8(j>>2) + 4*half -- the order in which the S accumulators
//     already hold the keys -- so P is only converted (v_cvt_pk_bf16_f32), never permuted.  The matching A operand needs V
//     TRANSPOSED: V tiles go global -> registers -> LDS as V^T[d][slot order], row stride 80 B (conflict-free b128 reads);
//   * the softmax scale is folded into the exponent (exp2(s*c - m)), not into bf16 Q.
// With the matrix work 16x cheaper than in fp32 this kernel is bound by the softmax VALU / transcendental work.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;       // head dim
constexpr int KT = 32;       // keys per tile
constexpr int VROW = 40;     // bf16 per V^T row in LDS: 32 slots + 8 padding (80-byte stride)
constexpr float RESCALE_THR = 16.f;

#define SCPB_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SCPB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// slot of `key` (0..31) in the k order of the P.V MFMAs (inverse of key = (j&3) + 16t + 8(j>>2) + 4h, slot = 16t + 8h + j)
__device__ __forceinline__ int vslot(int key) {
    return 16 * (key >> 4) + 8 * ((key >> 2) & 1) + (key & 3) + 4 * ((key >> 3) & 1);
}
__device__ __forceinline__ float other_half(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vit_attention_bf16_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ out,
                                                                       int N, int H, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) __bf16 k_lds[2][KT * HD];
    __shared__ __attribute__((aligned(16))) __bf16 v_lds[2][HD * VROW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int bh = blockIdx.y, qg = blockIdx.x;
    if ((gridDim.y & 7) == 0) {   // XCD-aware placement, as in the fp32 kernel
        const int lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int slot = lin >> 3;
        bh = (slot / gridDim.x) * 8 + (lin & 7);
        qg = slot % gridDim.x;
    }
    const int b = bh / H, h = bh - b * H;
    const int q0 = (qg * WAVES + wave) * 32;
    const size_t row_stride = (size_t)3 * H * HD;
    const __bf16* base = qkv + (size_t)b * N * row_stride + (size_t)h * HD;
    const __bf16* kbase = base + (size_t)H * HD;
    const __bf16* vbase = base + (size_t)2 * H * HD;
    const int ntiles = (N + KT - 1) / KT;

    // ---- K tile -> LDS by DMA: 4 pieces of 8 rows; lane L of a piece: row 8p + L/8, LDS slot L%8 <- chunk slot ^ ((row>>1)&7)
    auto issue_k = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < (4 + WAVES - 1) / WAVES; i++) {
            const int p = wave + WAVES * i;
            if (p >= 4) continue;
            const int r = 8 * p + (lane >> 3), slot = lane & 7;
            const int key = min(kt * KT + r, N - 1);
            const int chunk = slot ^ ((r >> 1) & 7);
            __builtin_amdgcn_global_load_lds(SCPB_GLOBAL_PTR(kbase + (size_t)key * row_stride + 8 * chunk),
                                             SCPB_LDS_PTR(k_lds[buf] + p * 512), 16, 0, 0);
        }
    };
    // ---- V tile: global -> registers (early) -> transposed LDS image (late); 256 chunks of 8 bf16 over the workgroup
    constexpr int VCH = (256 + WAVES * 64 - 1) / (WAVES * 64);
    bf16x8 vreg[VCH];
    auto load_v = [&](int kt) {
#pragma unroll
        for (int i = 0; i < VCH; i++) {
            const int c = tid + WAVES * 64 * i;
            if (c < 256) {
                const int key = min(kt * KT + (c >> 3), N - 1);
                vreg[i] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)key * row_stride + 8 * (c & 7));
            }
        }
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < VCH; i++) {
            const int c = tid + WAVES * 64 * i;
            if (c < 256) {
                const int s = vslot(c >> 3), d0 = 8 * (c & 7);
#pragma unroll
                for (int e = 0; e < 8; e++) v_lds[buf][(d0 + e) * VROW + s] = vreg[i][e];
            }
        }
    };

    // ---- Q fragments: qreg[ks] = Q[query = l31][16 ks + 8 half .. +7]
    bf16x8 qreg[4];
    {
        const int q = min(q0 + l31, N - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            qreg[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)q * row_stride + 16 * ks + 8 * half);
    }
    f32x16 o_lo, o_hi;
#pragma unroll
    for (int r = 0; r < 16; r++) { o_lo[r] = 0.f; o_hi[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    issue_k(0, 0);
    load_v(0);
    store_v(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < ntiles; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) {
            issue_k(kt + 1, buf ^ 1);
            load_v(kt + 1);
        }
        // ---- S^T = K Q^T (raw dot products; the softmax scale goes into the exponent)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.f;
        {
            const __bf16* krow = k_lds[buf] + l31 * HD;
            const int sw = (l31 >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const bf16x8 kk = *reinterpret_cast<const bf16x8*>(krow + 8 * ((2 * ks + half) ^ sw));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kk, qreg[ks], s, 0, 0, 0);
            }
        }
        const int key_base = kt * KT;
        if (key_base + KT > N) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (key_base + acc_row(r, half) >= N) s[r] = -INFINITY;
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) m_tile = fmaxf(m_tile, s[r]);
        m_tile = fmaxf(m_tile, other_half(m_tile)) * scale_log2e;
        if (__any(m_tile > m_run + RESCALE_THR)) {   // deferred rescale, as in the fp32 kernel
            const float m_new = fmaxf(m_run, m_tile);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; r++) { o_lo[r] *= alpha; o_hi[r] *= alpha; }
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2e, -m_run));
            psum += s[r];
        }
        l_run += psum;
        // ---- O^T += V^T P^T: two k-steps of 16 keys, two 32-row blocks of d
#pragma unroll
        for (int t = 0; t < 2; t++) {
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 8; j++) pb[j] = (__bf16)s[8 * t + j];
            const __bf16* vrow = v_lds[buf] + l31 * VROW + 16 * t + 8 * half;
            const bf16x8 va = *reinterpret_cast<const bf16x8*>(vrow);
            const bf16x8 vb = *reinterpret_cast<const bf16x8*>(vrow + 32 * VROW);
            o_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o_lo, 0, 0, 0);
            o_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pb, o_hi, 0, 0, 0);
        }
        if (kt + 1 < ntiles) store_v(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const float l_tot = l_run + other_half(l_run);
    const float inv = 1.f / l_tot;
    const int q = q0 + l31;
    if (q < N) {
        __bf16* op = out + ((size_t)b * N + q) * (H * HD) + (size_t)h * HD;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int d = acc_row(r, half);
            op[d] = (__bf16)(o_lo[r] * inv);
            op[d + 32] = (__bf16)(o_hi[r] * inv);
        }
    }
}

}  // namespace

extern "C" int scp_vit_attention_bf16_forward(const void* qkv, void* out, int B, int N, int H, int head_dim, float scale,
                                              void* stream) {
    if (B <= 0 || N <= 0 || H <= 0) return scp::fail(hipErrorInvalidValue, "vit_attention_bf16: empty problem");
    if (head_dim != HD) return scp::fail(hipErrorInvalidValue, "vit_attention_bf16: head_dim must be 64");
    if (!qkv || !out) return scp::fail(hipErrorInvalidValue, "vit_attention_bf16: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float sl = scale * 1.4426950408889634f;
    const int qtiles = (N + 31) / 32;
    hipLaunchKernelGGL(vit_attention_bf16_kernel<4>, dim3((qtiles + 3) / 4, B * H), dim3(256), 0, st,
                       static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), N, H, sl);
    return scp::check_launch("vit_attention_bf16");
}
