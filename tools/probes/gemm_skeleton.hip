// tools/probes/gemm_skeleton.hip -- which ingredient of the fp32 GEMM main loop costs matrix-pipe time?  One workgroup = 4
// wavefronts, 3 workgroups per CU, 32 MFMA 32x32x2 per chunk per wavefront (the 64 x 64 sub-tile of csrc/vit_gemm.hip), and
// per template flag: 1 = dependent order (4 consecutive MFMAs on one accumulator) instead of interleaved, 2 = the 8
// ds_read_b128 per chunk, 4 = s_barrier per chunk, 8 = the 4 LDS-DMA instructions per chunk (L2-resident source),
// 16 = DMA source streams from HBM-sized buffer, 32 = no vmcnt wait inside the loop, 64 = global_load to VGPRs + ds_write
// instead of LDS-DMA (waits for the registers one chunk later), 128 = 8 DMA instructions of 8 B... (unused).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gemm_skeleton.hip -o tools/probes/gemm_skeleton.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))

template <int FLAGS>
__global__ __launch_bounds__(256, 3) void skel(const float* src, unsigned src_floats, float* out, int chunks) {
    __shared__ __attribute__((aligned(16))) float a0[2048], a1[2048], a2[2048], w0[2048], w1[2048], w2[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    f32x4 av[2][2], wv[2][2];
    for (int i = 0; i < 2; i++) for (int c = 0; c < 2; c++) { av[i][c] = (f32x4){1.f + lane, 2.f, 3.f, 4.f}; wv[i][c] = (f32x4){0.5f, 0.25f, lane * 1.f, 1.f}; }
    if (!(FLAGS & 8)) { for (int i = tid; i < 2048; i += 256) { a0[i] = a1[i] = a2[i] = i; w0[i] = w1[i] = w2[i] = 1.f; } __syncthreads(); }
    const unsigned rd = (unsigned)((lane & 31) * 64 + (lane >> 5) * 32);      // bytes
    unsigned goff = (blockIdx.x * 4096u + wave * 512u + lane * 4u) & (unsigned)(src_floats - 1);
    const unsigned gstep = (FLAGS & 16) ? gridDim.x * 4096u : 0u;
    f32x4 stage_regs[4];
    unsigned gather_k = 0;
    auto issue = [&](float* ad, float* wd) {
        if (FLAGS & 64) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
                stage_regs[i] = *reinterpret_cast<const f32x4*>(src + ((goff + i * 256) & (src_floats - 1)));
                stage_regs[2 + i] = *reinterpret_cast<const f32x4*>(src + ((goff + 2048 + i * 256) & (src_floats - 1)));
            }
            goff = (goff + gstep) & (src_floats - 1);
            return;
        }
        if (FLAGS & 384) {
            // 128: the GEMM's gather -- one instruction = 16 rows x 64 B, row pitch 1536 B (K = 384), next chunk 64 B further
            // 256: 8 rows x 128 B per instruction (whole cache lines), next step 128 B further on alternating operands
            const int rpi = (FLAGS & 128) ? 16 : 8, lpr = 64 / rpi;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const unsigned row = rpi * (2 * wave + i) + lane / lpr, piece = lane % lpr;
                const unsigned oa = (blockIdx.x % 256u * 128u + row) * 384u + 4u * piece + gather_k;
                const unsigned ow = ((blockIdx.x / 256u) * 128u + row) * 384u + 4u * piece + gather_k + (1u << 26);
                __builtin_amdgcn_global_load_lds(GP(src + (oa & (src_floats - 1))), LP(ad + (2 * wave + i) * 256), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GP(src + (ow & (src_floats - 1))), LP(wd + (2 * wave + i) * 256), 16, 0, 0);
            }
            gather_k += (FLAGS & 128) ? 16 : 32;
            if (gather_k >= 384) gather_k = 0;
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            __builtin_amdgcn_global_load_lds(GP(src + ((goff + i * 256) & (src_floats - 1))), LP(ad + (2 * wave + i) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GP(src + ((goff + 2048 + i * 256) & (src_floats - 1))), LP(wd + (2 * wave + i) * 256), 16, 0, 0);
        }
        goff = (goff + gstep) & (src_floats - 1);
    };
    auto compute = [&](const float* as, const float* ws) {
        if (FLAGS & 2) {
            const unsigned ab = (unsigned)(size_t)LP(as) + rd, wb = (unsigned)(size_t)LP(ws) + rd;
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(av[i][c]) : "v"(ab + 16u * c + 4096u * i));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(wv[i][c]) : "v"(wb + 16u * c + 4096u * i));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0][0]), "+v"(av[0][1]), "+v"(av[1][0]), "+v"(av[1][1]), "+v"(wv[0][0]), "+v"(wv[0][1]), "+v"(wv[1][0]), "+v"(wv[1][1]));
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (FLAGS & 1) {
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int k = 0; k < 4; k++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c][k], wv[j][c][k], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c][k], wv[j][c][k], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto step = [&](int kc, const float* as, const float* ws, float* an, float* wn) {
        if ((FLAGS & 64) && kc >= 1) {      // registers loaded one chunk ago -> LDS
#pragma unroll
            for (int i = 0; i < 2; i++) {
                *reinterpret_cast<f32x4*>(an + (2 * wave + i) * 256 + lane * 4) = stage_regs[i];
                *reinterpret_cast<f32x4*>(wn + (2 * wave + i) * 256 + lane * 4) = stage_regs[2 + i];
            }
        }
        if ((FLAGS & 8) && !(FLAGS & 32) && !(FLAGS & 64)) {
            if (kc + 1 < chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (FLAGS & 4) __builtin_amdgcn_s_barrier();
        if ((FLAGS & 8) && kc + 2 < chunks) issue(an, wn);
        compute(as, ws);
    };
    if (FLAGS & 8) { issue(a0, w0); issue(a1, w1); }
    for (int kc = 0; kc < chunks; kc += 3) {
        step(kc, a0, w0, a2, w2);
        if (kc + 1 < chunks) step(kc + 1, a1, w1, a0, w0);
        if (kc + 2 < chunks) step(kc + 2, a2, w2, a1, w1);
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int FLAGS>
void run(const float* src, size_t n, float* out, int chunks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 768;
    hipLaunchKernelGGL(skel<FLAGS>, dim3(blocks), dim3(256), 0, 0, src, n, out, chunks); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(skel<FLAGS>, dim3(blocks), dim3(256), 0, 0, src, n, out, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * chunks * 32 * (2.0 * 32 * 32 * 2);
    printf("flags %2d (%s%s%s%s%s): %.1f TFLOP/s\n", FLAGS, FLAGS & 1 ? "dep " : "inter ", FLAGS & 2 ? "ldsread " : "", FLAGS & 4 ? "barrier " : "",
           FLAGS & 8 ? "dma " : "", FLAGS & 128 ? "gather16x64B " : FLAGS & 256 ? "gather8x128B " : FLAGS & 16 ? "stream " : "", flops / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t n = (size_t)1 << 28;      // 1 GiB of floats
    float *src, *out; hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4); hipMalloc(&out, 768 * 256 * 4);
    for (int c : {24, 72, 96, 288}) { printf("chunks %d: ", c); run<14 + 128>(src, n, out, c); }
    const int chunks = 24 * 12;
    run<0>(src, n, out, chunks); run<1>(src, n, out, chunks); run<2>(src, n, out, chunks); run<3>(src, n, out, chunks);
    run<4>(src, n, out, chunks); run<5>(src, n, out, chunks); run<6>(src, n, out, chunks); run<7>(src, n, out, chunks);
    run<14>(src, n, out, chunks); run<15>(src, n, out, chunks); run<30>(src, n, out, chunks); run<31>(src, n, out, chunks);
    run<8>(src, n, out, chunks); run<24>(src, n, out, chunks);
    run<14 + 128>(src, n, out, chunks); run<14 + 256>(src, n, out, chunks);
    run<8 + 32>(src, n, out, chunks); run<8 + 64>(src, n, out, chunks); run<8 + 64 + 4 + 2>(src, n, out, chunks); run<8 + 64 + 4 + 2 + 16>(src, n, out, chunks);
    return 0;
}
