// tools/probes/pk_forms.hip -- which INSTRUCTION of a victim kernel is disturbed by K-doubled 16-bit MFMAs on the same SIMD?  
// Round 4's one-ingredient victims were contracting maps (x <- f(x) with a fixed point): a transient wrong result was washed out before
// the store.  Here every result is checked INSIDE the kernel against the same product computed by the scalar VALU instruction, and the
// count of mismatches is what the kernel writes: one wrong lane in one iteration is seen.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/pk_forms.hip -o tools/probes/libpk_forms.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---- aggressor: register-only loop of v_mfma_f32_32x32x16_bf16 (kind 0) or v_mfma_f32_32x32x2_f32 (kind 1) -----------------------
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    bf16x8 a8, b8;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a8[i] = (__bf16)(seed * (float)(lane + i) * 1e-3f);
        b8[i] = (__bf16)(seed * (float)(lane - i) * 1e-3f);
    }
    const float fa = seed * lane * 1e-3f, fb = seed * (63 - lane) * 1e-3f;
    for (int k = 0; k < iters; k++) {
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc, 0, 0, 0);
        if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int pk_aggressor(int kind, float* out, int blocks, int iters, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f);
    else hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, st, out, iters, 1.f);
    return (int)hipGetLastError();
}

// ---- victims -----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ssub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t bits(float x) { return __float_as_uint(x); }

#define PK2(name, text)                                                                                                      \
    __device__ __forceinline__ f2 name(f2 a, f2 b) { f2 r; asm volatile(text : "=v"(r) : "v"(a), "v"(b)); return r; }
PK2(pk_mul_plain, "v_pk_mul_f32 %0, %1, %2")
PK2(pk_mul_x01, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")              // lo = a.lo * b.hi, hi = a.hi * b.lo
PK2(pk_mul_x10, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]")              // lo = a.hi * b.lo, hi = a.lo * b.hi
PK2(pk_mul_h10, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                           // lo = a.lo * b.lo, hi = a.hi * b.lo
PK2(pk_mul_h01, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                           // lo = a.lo * b.lo, hi = a.lo * b.hi
PK2(pk_add_neg, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")                 // a - b
PK2(pk_mul_x01_neg, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[0,1]")   // lo = a.lo * -b.hi, hi = a.lo * -b.lo
PK2(pk_mov_x10, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]")                              // lo = a.hi, hi = b.hi  (op_sel_hi default [1,1])
PK2(pk_mul_x11, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]")              // lo = a.hi * b.hi, hi = a.hi * b.lo
__device__ __forceinline__ f2 pk_fma_plain(f2 a, f2 b, f2 c) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f2 pk_fma_neg(f2 a, f2 b, f2 c) {
    f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int FORM>
__global__ __launch_bounds__(256) void victim(uint32_t* out, int iters, const float* mem) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float ax0 = 1.0f + (float)(i & 1023) * 9.765625e-4f, ay0 = 0.5f + (float)(i >> 10) * 1.220703125e-4f;
    const float bx0 = 1.25f + (float)(i & 511) * 1.953125e-3f, by0 = 0.75f + (float)(i >> 9) * 3.0517578125e-5f;
    uint32_t bad = 0;
    for (int k = 0; k < iters; k++) {
        const float t = (float)(k & 255) * 3.90625e-3f;
        f2 a = {ax0 + t, ay0 - t * 0.5f}, b = {bx0 - t * 0.25f, by0 + t};
        if (FORM == 20 || FORM == 21) {                    // operands straight from memory: the packed op is the first use after the wait
            const f2* m = reinterpret_cast<const f2*>(mem) + ((size_t)((i * 2 + k * 64) & 0xfffff));
            a = __builtin_nontemporal_load(m);
            b = __builtin_nontemporal_load(m + 1);
        }
        f2 r, e;
        if (FORM == 0) { r = pk_mul_plain(a, b); e = (f2){smul(a.x, b.x), smul(a.y, b.y)}; }
        if (FORM == 1 || FORM == 20) { r = pk_mul_x01(a, b); e = (f2){smul(a.x, b.y), smul(a.y, b.x)}; }
        if (FORM == 2 || FORM == 21) { r = pk_mul_x10(a, b); e = (f2){smul(a.y, b.x), smul(a.x, b.y)}; }
        if (FORM == 3) { r = pk_mul_h10(a, b); e = (f2){smul(a.x, b.x), smul(a.y, b.x)}; }
        if (FORM == 4) { r = pk_mul_h01(a, b); e = (f2){smul(a.x, b.x), smul(a.x, b.y)}; }
        if (FORM == 5) { r = pk_add_neg(a, b); e = (f2){ssub(a.x, b.x), ssub(a.y, b.y)}; }
        if (FORM == 6) { r = pk_fma_plain(a, b, a); e = (f2){sfma(a.x, b.x, a.x), sfma(a.y, b.y, a.y)}; }
        if (FORM == 7) { r = pk_mul_x01_neg(a, b); e = (f2){smul(a.x, -b.y), smul(a.x, -b.x)}; }
        if (FORM == 8) { r = pk_mov_x10(a, b); e = (f2){a.y, b.y}; }
        if (FORM == 9) { r = (f2){smul(a.x, b.y), smul(a.y, b.x)}; e = (f2){smul(a.x, b.y), smul(a.y, b.x)}; }        // control: scalar only
        if (FORM == 10) { r = pk_fma_neg(a, b, a); e = (f2){sfma(-a.x, b.x, a.x), sfma(-a.y, b.y, a.y)}; }
        if (FORM == 11) { r = pk_mul_x11(a, b); e = (f2){smul(a.y, b.y), smul(a.y, b.x)}; }
        if (FORM == 12) {
            // the opening of the SLP-built face_setup_kernel (the known victim; its element 2 = (x1 y2 - x2 y1) / det was the wrong one):
            // three packed products with op_sel, then scalar products written OVER the high halves of two of the packed results, then
            // a packed subtraction that reads the pairs.  Registers are named so that the pattern is the compiler's, byte for byte.
            float d2, d3, d23;
            asm volatile(
                "v_pk_mul_f32 v[12:13], %3, %4 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "v_pk_mul_f32 v[14:15], %3, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 v[10:11], %5, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                "v_mul_f32 v13, %6, %7\n\t"
                "v_mul_f32 v15, %8, %9\n\t"
                "v_pk_add_f32 v[22:23], v[10:11], v[10:11] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_add_f32 v[24:25], v[12:13], v[14:15] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_mov_b32 %0, v24\n\t"
                "v_mov_b32 %1, v25\n\t"
                "v_mov_b32 %2, v23"
                : "=v"(d2), "=v"(d3), "=v"(d23)
                : "v"(a), "v"(b), "v"((f2){b.y, a.x}), "v"(a.y), "v"(b.x), "v"(a.x), "v"(b.y)
                : "v10", "v11", "v12", "v13", "v14", "v15", "v22", "v23", "v24", "v25");
            // a = (x1,y1) in %3, b = (x2,y2) in %4, c = %5
            const float e2 = ssub(smul(a.x, b.y), smul(a.y, b.x));
            const float e3 = ssub(smul(a.y, b.x), smul(a.x, b.y));
            const f2 c = {b.y, a.x};
            const float e23 = ssub(smul(c.y, b.x), smul(c.x, b.y));
            bad += (bits(d2) != bits(e2)) + (bits(d3) != bits(e3)) + (bits(d23) != bits(e23));
            continue;
        }
        bad += (bits(r.x) != bits(e.x)) + (bits(r.y) != bits(e.y));
    }
    out[i] = bad;
}

#define LAUNCH(F) case F: hipLaunchKernelGGL(victim<F>, dim3(blocks), dim3(256), 0, st, out, iters, mem); break;
extern "C" int pk_victim(int form, uint32_t* out, int blocks, int iters, const float* mem, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (form) {
        LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10) LAUNCH(11) LAUNCH(12)
        LAUNCH(20) LAUNCH(21)
        default: return -1;
    }
    return (int)hipGetLastError();
}
