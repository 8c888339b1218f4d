// self-corr-pose_amd/csrc/vit_norm.hip -- fused residual-add + LayerNorm forward for the frozen DINO ViT.
//
// Replaces, per transformer block of third-party/zsp/zsp/method/vision_transformer_flexible.py:117-132,
//     x = x + branch;  y = LayerNorm(x)          (two element-wise passes + torch's LN kernel)
// by one HBM pass: read x (and the branch output), write the updated residual stream and its normalised
// copy.  HBM-bound (0 reuse): one wavefront per token row, the row (C <= 1024 floats) lives in registers,
// float2 accesses are coalesced (512 B per wavefront instruction), mean / variance by DPP-free butterfly
// shuffles.  LayerNorm semantics = torch.nn.LayerNorm(C, eps): biased variance, y = (x-mean)*rstd*g + b.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {

constexpr int LN_MAX_PAIRS = 8;   // 64 lanes x 8 float2 = 1024 columns

__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ branch,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, long rows,
                                                            int C, float* __restrict__ sum_out,
                                                            float* __restrict__ y_out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int pairs = C >> 1;                       // C is even
    const float2* xr = reinterpret_cast<const float2*>(x + (size_t)row * C);
    const float2* br = branch ? reinterpret_cast<const float2*>(branch + (size_t)row * C) : nullptr;
    float2 v[LN_MAX_PAIRS];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_PAIRS; j++) {
        const int p = lane + 64 * j;
        v[j] = make_float2(0.f, 0.f);
        if (p < pairs) {
            v[j] = xr[p];
            if (br) { const float2 b = br[p]; v[j].x += b.x; v[j].y += b.y; }
            s += v[j].x + v[j].y;
        }
    }
    const float mean = wave_allsum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_PAIRS; j++) {
        const int p = lane + 64 * j;
        if (p < pairs) {
            const float a = v[j].x - mean, b = v[j].y - mean;
            q += a * a + b * b;
        }
    }
    const float rstd = rsqrtf(wave_allsum(q) / C + eps);
    float2* so = sum_out ? reinterpret_cast<float2*>(sum_out + (size_t)row * C) : nullptr;
    float2* yo = reinterpret_cast<float2*>(y_out + (size_t)row * C);
    const float2* g2 = reinterpret_cast<const float2*>(gamma);
    const float2* b2 = reinterpret_cast<const float2*>(beta);
#pragma unroll
    for (int j = 0; j < LN_MAX_PAIRS; j++) {
        const int p = lane + 64 * j;
        if (p < pairs) {
            if (so) so[p] = v[j];
            const float2 g = g2[p], b = b2[p];
            yo[p] = make_float2((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y);
        }
    }
}

}  // namespace

extern "C" int scp_add_layernorm_forward(const float* x, const float* branch, const float* gamma,
                                         const float* beta, float eps, long rows, int C, float* sum_out,
                                         float* y_out, void* stream) {
    if (rows <= 0 || C <= 0) return scp::fail(hipErrorInvalidValue, "add_layernorm: empty problem");
    if (C % 2 != 0 || C > 128 * LN_MAX_PAIRS)
        return scp::fail(hipErrorInvalidValue, "add_layernorm: C must be even and <= 1024");
    hipLaunchKernelGGL(add_layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, branch, gamma, beta, eps, rows, C, sum_out, y_out);
    return scp::check_launch("add_layernorm");
}
