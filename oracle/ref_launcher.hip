// oracle/ref_launcher.hip -- TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Builds oracle/_ref/libref_softras*.so = the REFERENCE's own SoftRas kernels, compiled UNCHANGED by hipcc
// for gfx950.  The kernel text is not in this repository: oracle/build_ref.py extracts the anonymous
// namespace of
//   /root/reference/third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu  (lines 22-671:
//   device helpers + forward_soft_rasterize_inv / forward_soft_rasterize / backward_soft_rasterize kernels
//   on raw pointers)
// into a scratch file at build time and passes it as -DSCP_REF_KERNEL_BODY="...".  No stand-ins are
// needed: the body only uses __global__/__device__, threadIdx/blockIdx, atomicAdd and libm calls, all of which
// HIP provides natively.  What this file adds is the part of the reference that depends on ATen (the host
// launchers, kernel.cu:674-813), restated on plain pointers:
//   * grid = (n - 1) / 512 + 1 blocks of 512 threads           (kernel.cu:697-699, 714, 774-776)
//   * the face pre-pass then the pixel pass on the same stream  (kernel.cu:701-739)
//   * caller-zeroed / caller-initialised buffers, written in place (functional/soft_rasterize.py:47-53,88-89)
//   * float and double instantiations                           (AT_DISPATCH_FLOATING_TYPES, kernel.cu:701,716,779)
// texture_res = int(sqrt(texture_size)) as kernel.cu:696.
#include <hip/hip_runtime.h>
#include <math.h>

#ifndef SCP_REF_KERNEL_BODY
#error "build through oracle/build_ref.py (it extracts the reference kernel body at build time)"
#endif
#include SCP_REF_KERNEL_BODY

namespace {

template <typename T>
int ref_forward(const T* faces, const T* textures, T* faces_info, T* aggrs_info, T* soft_colors, int batch, int nfaces,
                int image_size, int texture_size, float near_, float far_, float eps, float sigma, int dist_mode,
                float dist_eps, float gamma, int rgb_mode, int alpha_mode, int sample_mode, int double_side,
                hipStream_t stream) {
    const int threads = 512;
    const int texture_res = int(sqrt((double)texture_size));
    const dim3 blocks_1((batch * nfaces - 1) / threads + 1);
    forward_soft_rasterize_inv_cuda_kernel<T><<<blocks_1, threads, 0, stream>>>(faces, faces_info, batch, nfaces, image_size);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
    const dim3 blocks_2((batch * image_size * image_size - 1) / threads + 1);
    forward_soft_rasterize_cuda_kernel<T><<<blocks_2, threads, 0, stream>>>(
        faces, textures, faces_info, aggrs_info, soft_colors, batch, nfaces, image_size, texture_size, texture_res, near_,
        far_, eps, sigma, dist_mode, dist_eps, gamma, rgb_mode, alpha_mode, sample_mode, (bool)double_side);
    return (int)hipGetLastError();
}

template <typename T>
int ref_backward(const T* faces, const T* textures, const T* soft_colors, const T* faces_info, const T* aggrs_info,
                 T* grad_faces, T* grad_textures, T* grad_soft_colors, int batch, int nfaces, int image_size,
                 int texture_size, float near_, float far_, float eps, float sigma, int dist_mode, float dist_eps,
                 float gamma, int rgb_mode, int alpha_mode, int sample_mode, int double_side, hipStream_t stream) {
    const int threads = 512;
    const int texture_res = int(sqrt((double)texture_size));
    const dim3 blocks((batch * image_size * image_size - 1) / threads + 1);
    backward_soft_rasterize_cuda_kernel<T><<<blocks, threads, 0, stream>>>(
        faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures, grad_soft_colors, batch, nfaces,
        image_size, texture_size, texture_res, near_, far_, eps, sigma, dist_mode, dist_eps, gamma, rgb_mode, alpha_mode,
        sample_mode, (bool)double_side);
    return (int)hipGetLastError();
}

}  // namespace

#define REF_FWD_ARGS(T)                                                                                                  \
    const T *faces, const T *textures, T *faces_info, T *aggrs_info, T *soft_colors, int batch, int nfaces,              \
        int image_size, int texture_size, float near_, float far_, float eps, float sigma, int dist_mode, float dist_eps, \
        float gamma, int rgb_mode, int alpha_mode, int sample_mode, int double_side, void *stream
#define REF_BWD_ARGS(T)                                                                                                  \
    const T *faces, const T *textures, const T *soft_colors, const T *faces_info, const T *aggrs_info, T *grad_faces,    \
        T *grad_textures, T *grad_soft_colors, int batch, int nfaces, int image_size, int texture_size, float near_,     \
        float far_, float eps, float sigma, int dist_mode, float dist_eps, float gamma, int rgb_mode, int alpha_mode,    \
        int sample_mode, int double_side, void *stream

extern "C" {
int ref_soft_rasterize_forward_f32(REF_FWD_ARGS(float)) {
    return ref_forward<float>(faces, textures, faces_info, aggrs_info, soft_colors, batch, nfaces, image_size, texture_size,
                              near_, far_, eps, sigma, dist_mode, dist_eps, gamma, rgb_mode, alpha_mode, sample_mode,
                              double_side, (hipStream_t)stream);
}
int ref_soft_rasterize_forward_f64(REF_FWD_ARGS(double)) {
    return ref_forward<double>(faces, textures, faces_info, aggrs_info, soft_colors, batch, nfaces, image_size, texture_size,
                               near_, far_, eps, sigma, dist_mode, dist_eps, gamma, rgb_mode, alpha_mode, sample_mode,
                               double_side, (hipStream_t)stream);
}
int ref_soft_rasterize_backward_f32(REF_BWD_ARGS(float)) {
    return ref_backward<float>(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
                               grad_soft_colors, batch, nfaces, image_size, texture_size, near_, far_, eps, sigma, dist_mode,
                               dist_eps, gamma, rgb_mode, alpha_mode, sample_mode, double_side, (hipStream_t)stream);
}
int ref_soft_rasterize_backward_f64(REF_BWD_ARGS(double)) {
    return ref_backward<double>(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
                                grad_soft_colors, batch, nfaces, image_size, texture_size, near_, far_, eps, sigma, dist_mode,
                                dist_eps, gamma, rgb_mode, alpha_mode, sample_mode, double_side, (hipStream_t)stream);
}
/* 1 = built with hipcc's default -ffp-contract=fast (what nvcc does to the authors' build), 0 = -ffp-contract=off */
int ref_soft_rasterize_contracted(void) {
#ifdef SCP_REF_CONTRACT_OFF
    return 0;
#else
    return 1;
#endif
}
}
