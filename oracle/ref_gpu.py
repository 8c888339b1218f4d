"""oracle/ref_gpu.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end of oracle/_ref/libref_softras*.so: the REFERENCE's own SoftRas kernels
(/root/reference/third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu:22-671, compiled unchanged by
hipcc for gfx950 through oracle/build_ref.py + oracle/ref_launcher.hip), driven with the reference's launch
geometry.  Works on torch CUDA(=HIP) tensors, fp32 or fp64 (AT_DISPATCH_FLOATING_TYPES).  Only tests/,
__graft_entry__.smoke() and bench.py's reference-timing leg may import this module.

`render()` follows SoftRasterizeFunction.forward/backward
(/root/reference/third-party/softras/soft_renderer/functional/soft_rasterize.py:12-102): same buffer initialisation,
same scalar mapping.
"""
import ctypes
import math
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
VARIANTS = {"contract": "libref_softras.so", "nocontract": "libref_softras_nocontract.so"}

DIST = {"hard": 0, "barycentric": 1, "euclidean": 2}
RGB = {"hard": 0, "softmax": 1}
ALPHA = {"hard": 0, "sum": 1, "prod": 2}
SAMPLE = {"surface": 0, "vertex": 1}

_libs = {}


def available():
    return all(os.path.exists(os.path.join(_DIR, n)) for n in VARIANTS.values())


def lib(variant="nocontract"):
    if variant not in _libs:
        path = os.path.join(_DIR, VARIANTS[variant])
        if not os.path.exists(path):
            from oracle import build_ref
            build_ref.build()
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref is not built (needs /root/reference + hipcc: python oracle/build_ref.py)")
        L = ctypes.CDLL(path)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        scal = [ci, ci, ci, ci, cf, cf, cf, cf, ci, cf, cf, ci, ci, ci, ci, vp]
        for sfx in ("f32", "f64"):
            f = getattr(L, "ref_soft_rasterize_forward_" + sfx)
            f.argtypes = [vp] * 5 + scal
            f.restype = ci
            b = getattr(L, "ref_soft_rasterize_backward_" + sfx)
            b.argtypes = [vp] * 8 + scal
            b.restype = ci
        L.ref_soft_rasterize_contracted.restype = ci
        assert L.ref_soft_rasterize_contracted() == (1 if variant == "contract" else 0)
        _libs[variant] = L
    return _libs[variant]


def _sfx(t):
    import torch
    assert t.is_cuda and t.is_contiguous()
    return {torch.float32: "f32", torch.float64: "f64"}[t.dtype]


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def forward(faces, textures, faces_info, aggrs_info, soft_colors, image_size, near, far, eps, sigma_val, func_id_dist,
            dist_eps, gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side, variant="nocontract"):
    """argument list of the reference's forward_soft_rasterize (soft_rasterize_cuda.cpp:59-91); in place"""
    fn = getattr(lib(variant), "ref_soft_rasterize_forward_" + _sfx(faces))
    B, F = faces.shape[0], faces.shape[1]
    err = fn(faces.data_ptr(), textures.data_ptr(), faces_info.data_ptr(), aggrs_info.data_ptr(), soft_colors.data_ptr(),
             B, F, image_size, textures.shape[2], near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb,
             func_id_alpha, texture_sample_type, int(bool(double_side)), _stream())
    if err:
        raise RuntimeError("reference forward launch failed: hipError %d" % err)
    return faces_info, aggrs_info, soft_colors


def backward(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures, grad_soft_colors, image_size,
             near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha, texture_sample_type,
             double_side, variant="nocontract"):
    """argument list of the reference's backward_soft_rasterize (soft_rasterize_cuda.cpp:94-132); in place"""
    fn = getattr(lib(variant), "ref_soft_rasterize_backward_" + _sfx(faces))
    B, F = faces.shape[0], faces.shape[1]
    err = fn(faces.data_ptr(), textures.data_ptr(), soft_colors.data_ptr(), faces_info.data_ptr(), aggrs_info.data_ptr(),
             grad_faces.data_ptr(), grad_textures.data_ptr(), grad_soft_colors.data_ptr(), B, F, image_size,
             textures.shape[2], near, far, eps, sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
             texture_sample_type, int(bool(double_side)), _stream())
    if err:
        raise RuntimeError("reference backward launch failed: hipError %d" % err)
    return grad_faces, grad_textures


def scalars(image_size=256, near=1., far=100., fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func="euclidean",
            dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", aggr_func_alpha="prod", texture_type="surface"):
    return (int(image_size), float(near), float(far), float(eps), float(sigma_val), DIST[dist_func],
            float(math.log(1. / dist_eps - 1.)), float(gamma_val), RGB[aggr_func_rgb], ALPHA[aggr_func_alpha],
            SAMPLE[texture_type], bool(fill_back))


def render(face_vertices, face_textures, image_size=256, background_color=(0, 0, 0), grad_soft_colors=None,
           variant="nocontract", dtype=None, device="cuda", **kw):
    """numpy/torch in, dict of numpy arrays out (soft_colors, aggrs_info, faces_info [, grad_faces, grad_textures])"""
    import torch
    dtype = dtype or torch.float32
    fv = torch.as_tensor(face_vertices).to(device=device, dtype=dtype)
    B = fv.shape[0]
    fv = fv.reshape(B, -1, 9).contiguous()
    F = fv.shape[1]
    tex = torch.as_tensor(face_textures).to(device=device, dtype=dtype).reshape(B, F, -1, 3).contiguous()
    info = torch.zeros(B, F, 27, device=device, dtype=dtype)
    aggr = torch.zeros(B, 2, image_size, image_size, device=device, dtype=dtype)
    col = torch.ones(B, 4, image_size, image_size, device=device, dtype=dtype)
    for k in range(3):
        col[:, k] *= background_color[k]
    scal = scalars(image_size=image_size, **kw)
    forward(fv, tex, info, aggr, col, *scal, variant=variant)
    out = {"soft_colors": col, "aggrs_info": aggr, "faces_info": info}
    if grad_soft_colors is not None:
        g = torch.as_tensor(grad_soft_colors).to(device=device, dtype=dtype).contiguous()
        gf = torch.zeros_like(fv)
        gt = torch.zeros_like(tex)
        backward(fv, tex, col, info, aggr, gf, gt, g, *scal, variant=variant)
        out["grad_faces"] = gf
        out["grad_textures"] = gt
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}
