// tests/golden/ref_shim.cpp -- fixture GENERATOR support (build container only).
// Host launchers that replay the reference's <<<grid, 512>>> launches
// (soft_rasterize_cuda_kernel.cu:698-739, :775-806) sequentially over the kernel bodies that
// ref_harness.py extracted into kernel_body.inc at run time.
#include "ref_shim.hpp"
#include "kernel_body.inc"

namespace {
template <class K> void replay(long n, K&& kernel) {
    blockDim.x = 512;
    const long blocks = (n - 1) / 512 + 1;
    for (long b = 0; b < blocks; b++) {
        blockIdx.x = (int)b;
        for (int t = 0; t < 512; t++) { threadIdx.x = t; kernel(); }
    }
}
}  // namespace

extern "C" void ref_forward(const float* faces, const float* textures, float* faces_info,
                            float* aggrs_info, float* soft_colors, int B, int F, int S, int T, int R,
                            float near_, float far_, float eps, float sigma, int dist_id,
                            float dist_eps, float gamma, int rgb_id, int alpha_id, int sample_id,
                            int double_side) {
    replay((long)B * F, [&] {
        forward_soft_rasterize_inv_cuda_kernel<float>(faces, faces_info, B, F, S);
    });
    replay((long)B * S * S, [&] {
        forward_soft_rasterize_cuda_kernel<float>(faces, textures, faces_info, aggrs_info,
                                                  soft_colors, B, F, S, T, R, near_, far_, eps, sigma,
                                                  dist_id, dist_eps, gamma, rgb_id, alpha_id,
                                                  sample_id, double_side != 0);
    });
}

extern "C" void ref_backward(const float* faces, const float* textures, const float* soft_colors,
                             const float* faces_info, const float* aggrs_info, float* grad_faces,
                             float* grad_textures, float* grad_soft_colors, int B, int F, int S,
                             int T, int R, float near_, float far_, float eps, float sigma,
                             int dist_id, float dist_eps, float gamma, int rgb_id, int alpha_id,
                             int sample_id, int double_side) {
    replay((long)B * S * S, [&] {
        backward_soft_rasterize_cuda_kernel<float>(faces, textures, soft_colors, faces_info,
                                                   aggrs_info, grad_faces, grad_textures,
                                                   grad_soft_colors, B, F, S, T, R, near_, far_, eps,
                                                   sigma, dist_id, dist_eps, gamma, rgb_id, alpha_id,
                                                   sample_id, double_side != 0);
    });
}
