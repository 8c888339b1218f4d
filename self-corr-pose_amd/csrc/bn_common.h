// self-corr-pose_amd/csrc/bn_common.h -- pieces of the BatchNorm kernels (csrc/batchnorm.hip) that the convolution epilogue
// (csrc/conv_igemm.hip) shares: the per-(block, channel) partial-sum protocol -- write-through partial stores, an atomic
// ticket, the LAST workgroup of the launch folds the partials in fp64 in a fixed order and finalises the batch statistics --
// so that a convolution can hand its output's statistics to the BatchNorm that follows without a pass over the activation.
// Reference semantics: nn.BatchNorm2d as used by torchvision's BasicBlock (model/module/network/image_encoder.py:119-139).
#pragma once
#include <hip/hip_runtime.h>

namespace scp_bn {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_BLOCKS = 1024;

struct Geometry {
    int tc;              // threads across channels = C/4
    int ri;              // row lanes per block = 256 / tc
    int rows_per_block;  // multiple of ri
    int blocks;
};

inline Geometry geometry(long R, int C) {
    Geometry g;
    g.tc = C / 4;
    g.ri = BN_THREADS / g.tc;
    long rows = (long)g.ri * 16;  // >= 16 float4 loads per thread
    long blocks = (R + rows - 1) / rows;
    if (blocks > BN_MAX_BLOCKS) {
        rows = ((R + BN_MAX_BLOCKS - 1) / BN_MAX_BLOCKS + g.ri - 1) / g.ri * g.ri;
        blocks = (R + rows - 1) / rows;
    }
    g.rows_per_block = (int)rows;
    g.blocks = (int)blocks;
    return g;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// bf16 storage (BASELINE configs[4] precision): 4 values = 8 bytes, arithmetic and statistics stay fp32
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const __bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ float bn_apply(float x, float scale, float shift) { return fmaf(x, scale, shift); }

// write-through stores (sc1): visible to the other XCDs once complete, without a release fence -- a release fence at agent
// scope writes back the WHOLE L2 of the XCD (buffer_wbl2), per workgroup, which cost more than the launch it was meant to save
__device__ __forceinline__ void st4_agent(float* p, float4 v) {
    __hip_atomic_store(p + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// folds the row lanes of a block in a fixed order and stores the block's partial for 4 channels
__device__ __forceinline__ void fold_rows(float4 a, float4 b, int tc_n, int ri_n, float* pa, float* pb, int C,
                                          float4* lds) {
    const int tc = threadIdx.x % tc_n, ri = threadIdx.x / tc_n;
    lds[threadIdx.x] = a;
    lds[BN_THREADS + threadIdx.x] = b;
    __syncthreads();
    if (ri == 0) {
        float4 sa = lds[tc], sb = lds[BN_THREADS + tc];
        for (int k = 1; k < ri_n; k++) {
            const float4 u = lds[k * tc_n + tc], v = lds[BN_THREADS + k * tc_n + tc];
            sa.x += u.x; sa.y += u.y; sa.z += u.z; sa.w += u.w;
            sb.x += v.x; sb.y += v.y; sb.z += v.z; sb.w += v.w;
        }
        st4_agent(pa + (size_t)blockIdx.x * C + 4 * tc, sa);
        st4_agent(pb + (size_t)blockIdx.x * C + 4 * tc, sb);
    }
}

// ---- last-block finalisation.  The per-(block, channel) fp32 partials are folded in fp64 by whichever workgroup of the
// SAME launch arrives last at a ticket counter (write-through partials -> atomic ticket -> acquire fence), not by a second tiny
// launch: inside the training step such a launch (<= 256 workgroups of work for a few microseconds) waited 37-57 us in the
// dispatcher behind the other streams' kernels, 84 times per step.  The fold order depends only on (blocks, C), never on
// which workgroup happens to be last, so results stay deterministic.  The ticket word is zero on entry and is re-armed
// (zeroed) by the last workgroup.
// `flag`: one LDS word of the caller's own buffer that is free at this point (the convolution kernels pass the first word of
// their operand ring: a separate __shared__ word pushed their 40 960-byte ring over a quarter of the CU's 160 KiB and cost the
// fourth resident workgroup -- round 4).
__device__ __forceinline__ bool last_block_arrived(unsigned* ticket, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's write-through partial stores have completed ...
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the ticket
        const int last = t == gridDim.x - 1;
        *flag = last;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int is_last = *flag;
    __syncthreads();                                     // everyone has read the flag before the caller reuses the word
    if (!is_last) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one workgroup: drop what this XCD's L2 may hold of the partial buffers
    return true;
}
__device__ __forceinline__ bool last_block_arrived(unsigned* ticket) {
    __shared__ int is_last;
    return last_block_arrived(ticket, &is_last);
}

// all BN_THREADS threads: thread (tc, ri) strides over the blocks k = ri, ri + ri_n, ... with up to 16 independent float4
// loads in flight, accumulates in fp64, then the ri_n lanes of a channel quad are added in lane order.  Returns true in the
// threads (ri == 0) that hold the totals of channels 4*tc .. 4*tc+3.
__device__ __forceinline__ bool fold_partials(const float* __restrict__ pa, const float* __restrict__ pb, int blocks, int C,
                                              int tc_n, double (&sa)[4], double (&sb)[4], float4* lds) {
    // the caller's 8 KB row-fold buffer is reused as 256 x 4 doubles, once per quantity: these kernels run beside LDS-heavy
    // kernels of the other streams and only get on a CU while their own LDS request fits into what is left
    double(*fl)[4] = reinterpret_cast<double(*)[4]>(lds);
    const int ri_n = BN_THREADS / tc_n;
    const int tc = threadIdx.x % tc_n, ri = threadIdx.x / tc_n;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = ri; k0 < blocks; k0 += 8 * ri_n) {
        float4 u[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int k = k0 + j * ri_n;
            const size_t o = (size_t)(k < blocks ? k : k0) * C + 4 * tc;
            u[j] = ld4(pa + o);
            v[j] = ld4(pb + o);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (k0 + j * ri_n < blocks) {
                a[0] += (double)u[j].x; a[1] += (double)u[j].y; a[2] += (double)u[j].z; a[3] += (double)u[j].w;
                b[0] += (double)v[j].x; b[1] += (double)v[j].y; b[2] += (double)v[j].z; b[3] += (double)v[j].w;
            }
        }
    }
    __syncthreads();            // the row fold's reads of `lds` are complete in every wavefront
#pragma unroll
    for (int i = 0; i < 4; i++) fl[threadIdx.x][i] = a[i];
    __syncthreads();
    if (ri == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) sa[i] = 0.0;
        for (int k = 0; k < ri_n; k++)
#pragma unroll
            for (int i = 0; i < 4; i++) sa[i] += fl[k * tc_n + tc][i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) fl[threadIdx.x][i] = b[i];
    __syncthreads();
    if (ri != 0) return false;
#pragma unroll
    for (int i = 0; i < 4; i++) sb[i] = 0.0;
    for (int k = 0; k < ri_n; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) sb[i] += fl[k * tc_n + tc][i];
    return true;
}

struct FwdFinalize {
    long R;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long* batches_tracked;
    float momentum, eps;
    float *mean_out, *invstd_out, *scale_out, *shift_out;
};

// per channel: batch statistics -> (mean, invstd, scale, shift) and the running-stat update
__device__ __forceinline__ void finalize_channel(const FwdFinalize& f, int c, bool training, double s, double q) {
    float mean, invstd;
    if (training) {
        const double m = s / (double)f.R;
        double var = q / (double)f.R - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
        if (f.running_var) {
            const double unbiased = f.R > 1 ? var * ((double)f.R / (double)(f.R - 1)) : var;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
        }
    } else {
        mean = f.running_mean[c];
        invstd = 1.f / sqrtf(f.running_var[c] + f.eps);
    }
    const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
    const float scale = g * invstd;
    f.mean_out[c] = mean;
    f.invstd_out[c] = invstd;
    f.scale_out[c] = scale;
    f.shift_out[c] = b - mean * scale;
}


}  // namespace scp_bn
