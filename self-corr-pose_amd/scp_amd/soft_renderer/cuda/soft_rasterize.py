"""scp_amd.soft_renderer.cuda.soft_rasterize -- the native boundary with the REFERENCE SIGNATURES.

Replaces the pybind module of third-party/softras/soft_renderer/cuda/soft_rasterize_cuda.cpp
(:59-91 forward_soft_rasterize, :94-132 backward_soft_rasterize, :135-138 module def): same
positional arguments, same in-place/return behaviour (caller allocates and pre-initialises every
buffer, the same tensors are returned), same RuntimeError on CPU / non-contiguous tensors.  The
work is done by the HIP kernels in csrc/softras.hip (float) / csrc/softras_f64.hip (double) through the C ABI of include/scp_hip.h, on
torch's current stream (the reference launches on the legacy default stream).
Difference: launch failures raise instead of being printf'd (kernel.cu:710-712).
"""
from ... import capi


def _params(faces, textures, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
            gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side):
    return capi.RasterParams(faces.shape[0], faces.shape[1], int(image_size), textures.shape[2],
                             float(near), float(far), float(eps), float(sigma_val), int(func_id_dist),
                             float(dist_eps), float(gamma_val), int(func_id_rgb), int(func_id_alpha),
                             int(texture_sample_type), int(bool(double_side)))


def _is_double(t):
    """AT_DISPATCH_FLOATING_TYPES (soft_rasterize_cuda_kernel.cu:701): float -> the tuned kernels, double -> csrc/softras_f64.hip"""
    import torch
    return t.dtype == torch.float64


def forward_soft_rasterize(faces, textures, faces_info, aggrs_info, soft_colors, image_size, near,
                           far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb,
                           func_id_alpha, texture_sample_type, double_side):
    p = _params(faces, textures, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
                gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side)
    f64 = _is_double(faces)
    ptr = capi.dev_ptr64 if f64 else capi.dev_ptr
    fn = capi.lib().scp_soft_rasterize_forward_f64 if f64 else capi.lib().scp_soft_rasterize_forward
    code = fn(ptr(faces, "faces"), ptr(textures, "textures"), ptr(faces_info, "faces_info"), ptr(aggrs_info, "aggrs_info"),
              ptr(soft_colors, "soft_colors"), p, capi.current_stream())
    capi.check(code, "scp_soft_rasterize_forward")
    return [faces_info, aggrs_info, soft_colors]


def forward_soft_rasterize_dual(faces, textures, faces_info, aggrs_info, soft_colors, textures_hard, aggrs_info_hard,
                                soft_colors_hard, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
                                gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side):
    """two forward_soft_rasterize calls that share their coverage (softmax-rgb primary + hard-rgb secondary on the same
    projected faces, same sigma / dist / alpha functions) as ONE launch: include/scp_hip.h scp_soft_rasterize_forward_dual.
    The scalars are the primary pass' (the reference's argument order); buffers follow the reference's protocol."""
    p = _params(faces, textures, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
                gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side)
    code = capi.lib().scp_soft_rasterize_forward_dual(
        capi.dev_ptr(faces, "faces"), capi.dev_ptr(textures, "textures"),
        capi.dev_ptr(faces_info, "faces_info"), capi.dev_ptr(aggrs_info, "aggrs_info"),
        capi.dev_ptr(soft_colors, "soft_colors"), capi.dev_ptr(textures_hard, "textures_hard"),
        capi.dev_ptr(aggrs_info_hard, "aggrs_info_hard"), capi.dev_ptr(soft_colors_hard, "soft_colors_hard"),
        p, capi.current_stream())
    capi.check(code, "scp_soft_rasterize_forward_dual")
    return [faces_info, aggrs_info, soft_colors, aggrs_info_hard, soft_colors_hard]


def backward_soft_rasterize(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces,
                            grad_textures, grad_soft_colors, image_size, near, far, eps, sigma_val,
                            func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
                            texture_sample_type, double_side):
    p = _params(faces, textures, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
                gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side)
    f64 = _is_double(faces)
    ptr = capi.dev_ptr64 if f64 else capi.dev_ptr
    fn = capi.lib().scp_soft_rasterize_backward_f64 if f64 else capi.lib().scp_soft_rasterize_backward
    code = fn(ptr(faces, "faces"), ptr(textures, "textures"), ptr(soft_colors, "soft_colors"), ptr(faces_info, "faces_info"),
              ptr(aggrs_info, "aggrs_info"), ptr(grad_faces, "grad_faces"), ptr(grad_textures, "grad_textures"),
              ptr(grad_soft_colors, "grad_soft_colors"), p, capi.current_stream())
    capi.check(code, "scp_soft_rasterize_backward")
    return [grad_faces, grad_textures]


def count_pairs(faces, image_size, sigma_val, dist_eps):
    """instrumentation (no reference counterpart): bbox-surviving (pixel, face) pairs of one pass"""
    import torch
    count = torch.zeros(1, dtype=torch.int64, device=faces.device)
    p = capi.RasterParams(faces.shape[0], faces.shape[1], int(image_size), 1, 1., 100., 1e-3,
                          float(sigma_val), 2, float(dist_eps), 1e-4, 0, 2, 0, 1)
    import ctypes
    code = capi.lib().scp_soft_rasterize_count_pairs(capi.dev_ptr(faces, "faces"),
                                                     ctypes.c_void_p(count.data_ptr()), p,
                                                     capi.current_stream())
    capi.check(code, "scp_soft_rasterize_count_pairs")
    return int(count.item())
