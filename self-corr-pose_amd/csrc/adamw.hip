// self-corr-pose_amd/csrc/adamw.hip -- the AdamW update of model/module/optimizers.py:77-79 (torch.optim.AdamW, decoupled weight decay) for
// ALL trainable parameters in ONE launch, on the flat gradient buffer of scp_amd.parallel.FlatGradients.
//
// torch's fused AdamW (multi_tensor_apply) takes 5 launches and 0.43 ms per step for the 14.7 M parameters here -- 8x the time the
// 7 streams of 58 MB (read p, g, m, v; write p, m, v) need at HBM speed -- on the step's critical path (nothing else runs behind the
// backward).  Here: parameter storages stay where they are (a table of device pointers), gradients and both moments are flat buffers
// in the same order (a parameter's flat segment is its storage order: FlatGradients' views carry the parameter's own strides), and a
// chunk list built once on the host deals 4096-element chunks to workgroups.  Per element, torch's formulas in torch's order
// (aten/src/ATen/native/cuda/fused_adam_utils.cuh, ADAMW, no amsgrad, no maximize):
//     p -= lr * wd * p;  m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) * g * g;
//     p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// with per-tensor lr / bias corrections (a tensor that starts receiving gradients later has its own step count, like torch's per-
// parameter `step`).  Inactive tensors (grad = None this step) are skipped.  HBM-bound: 28 B per parameter.
// The per-step scalars travel as KERNEL ARGUMENTS (scp_adamw_step, one row per class of tensors that share parameter group and step
// count); the device table is static.  The first version uploaded a per-tensor table every step through a ring of pinned staging
// buffers: the only host->device copy on the optimizer's path, and the one ingredient torch's own AdamW does not have.
#include <hip/hip_runtime.h>

#include "scp_common.h"
#include "scp_hip.h"

namespace {
constexpr int CHUNK = SCP_ADAMW_CHUNK;

__global__ __launch_bounds__(256) void adamw_flat_kernel(const scp_adamw_tensor* __restrict__ table, const int2* __restrict__ chunks,
                                                         const float* __restrict__ grad, float* __restrict__ exp_avg,
                                                         float* __restrict__ exp_avg_sq, const scp_adamw_step step, float beta1,
                                                         float beta2, float eps) {
    const int2 c = chunks[blockIdx.x];                   // (tensor index, first element of the chunk)
    const scp_adamw_tensor t = table[c.x];
    if (t.cls < 0) return;
    float* __restrict__ p = reinterpret_cast<float*>(t.param);
    const long long n = t.numel;
    const long long base = t.flat_offset;
    const float lr_wd = step.lr_wd[t.cls], step_size = step.step_size[t.cls], inv_bc2_sqrt = step.inv_bias_correction2_sqrt[t.cls];
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    const long long j0 = (long long)c.y, j1 = j0 + CHUNK < n ? j0 + CHUNK : n;
    // 4 elements per thread and iteration where the segment is 16-byte aligned on both sides
    const bool vec = (((size_t)p | (size_t)(grad + base) | (size_t)(exp_avg + base) | (size_t)(exp_avg_sq + base)) & 15) == 0 && (j0 & 3) == 0;
    long long j = j0 + (vec ? 4LL * threadIdx.x : (long long)threadIdx.x);
    auto update = [&](float& pv, float g, float& m, float& v) {
        pv -= lr_wd * pv;
        m = m + omb1 * (g - m);
        v = beta2 * v + omb2 * g * g;
        const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
        pv -= step_size * m / denom;
    };
    if (vec) {
        for (; j + 3 < j1; j += 4 * 256) {
            float4 pv = *reinterpret_cast<const float4*>(p + j);
            const float4 g = *reinterpret_cast<const float4*>(grad + base + j);
            float4 m = *reinterpret_cast<const float4*>(exp_avg + base + j);
            float4 v = *reinterpret_cast<const float4*>(exp_avg_sq + base + j);
            update(pv.x, g.x, m.x, v.x); update(pv.y, g.y, m.y, v.y); update(pv.z, g.z, m.z, v.z); update(pv.w, g.w, m.w, v.w);
            *reinterpret_cast<float4*>(p + j) = pv;
            *reinterpret_cast<float4*>(exp_avg + base + j) = m;
            *reinterpret_cast<float4*>(exp_avg_sq + base + j) = v;
        }
        // tail of the chunk (fewer than 4 left for this lane's slot): the first lanes take one element each
        const long long tail0 = j0 + ((j1 - j0) & ~3LL);
        j = tail0 + threadIdx.x;
        if (j < j1) {
            float pv = p[j], m = exp_avg[base + j], v = exp_avg_sq[base + j];
            update(pv, grad[base + j], m, v);
            p[j] = pv; exp_avg[base + j] = m; exp_avg_sq[base + j] = v;
        }
    } else {
        for (; j < j1; j += 256) {
            float pv = p[j], m = exp_avg[base + j], v = exp_avg_sq[base + j];
            update(pv, grad[base + j], m, v);
            p[j] = pv; exp_avg[base + j] = m; exp_avg_sq[base + j] = v;
        }
    }
}
}  // namespace

extern "C" int scp_adamw_flat(const scp_adamw_tensor* table, const int* chunks, int nchunks, const float* grad, float* exp_avg,
                              float* exp_avg_sq, const scp_adamw_step* step, float beta1, float beta2, float eps, void* stream) {
    if (nchunks <= 0) return 0;
    if (!step) return scp::fail(hipErrorInvalidValue, "scp_adamw_flat: step scalars missing");
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream), table,
                       reinterpret_cast<const int2*>(chunks), grad, exp_avg, exp_avg_sq, *step, beta1, beta2, eps);
    return scp::check_launch("adamw_flat");
}
