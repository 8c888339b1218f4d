"""oracle/softras.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

ctypes front-end of oracle/liboracle_softras.so (the plain-C restatement of the reference SoftRas
kernels, see softras_oracle.c for the file:line map).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.

The calling convention mirrors the reference's native boundary
(/root/reference/third-party/softras/soft_renderer/cuda/soft_rasterize_cuda.cpp:59-132):
the caller allocates and pre-initialises every output, the callee writes in place.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_softras.so")

DIST = {"hard": 0, "barycentric": 1, "euclidean": 2}
RGB = {"hard": 0, "softmax": 1}
ALPHA = {"hard": 0, "sum": 1, "prod": 2}
SAMPLE = {"surface": 0, "vertex": 1}


class _Params(ctypes.Structure):
    _fields_ = [
        ("batch", ctypes.c_int), ("nfaces", ctypes.c_int), ("size", ctypes.c_int),
        ("tex_size", ctypes.c_int), ("tex_res", ctypes.c_int),
        ("near_", ctypes.c_float), ("far_", ctypes.c_float), ("eps", ctypes.c_float),
        ("sigma", ctypes.c_float), ("dist_eps", ctypes.c_float), ("gamma", ctypes.c_float),
        ("dist_mode", ctypes.c_int), ("rgb_mode", ctypes.c_int), ("alpha_mode", ctypes.c_int),
        ("sample_mode", ctypes.c_int), ("double_side", ctypes.c_int),
    ]


def build(force=False):
    """(Re)build the oracle with gcc; building the checker is not using it."""
    src = os.path.join(_HERE, "softras_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle_softras.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        pp = ctypes.POINTER(_Params)
        _lib.sr_oracle_forward.argtypes = [fp, fp, fp, fp, fp, pp]
        _lib.sr_oracle_forward.restype = None
        _lib.sr_oracle_backward.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, pp]
        _lib.sr_oracle_backward.restype = None
        _lib.sr_oracle_count_pairs.argtypes = [fp, pp]
        _lib.sr_oracle_count_pairs.restype = ctypes.c_long
    return _lib


def _ptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _params(faces, textures, image_size, near, far, eps, sigma_val, dist_eps_log, gamma_val,
            func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side):
    B, F = faces.shape[:2]
    T = textures.shape[2]
    return _Params(B, F, image_size, T, int(np.sqrt(T)), near, far, eps, sigma_val, dist_eps_log,
                   gamma_val, func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type,
                   int(bool(double_side)))


def forward(faces, textures, faces_info, aggrs_info, soft_colors, image_size, near, far, eps,
            sigma_val, func_id_dist, dist_eps, gamma_val, func_id_rgb, func_id_alpha,
            texture_sample_type, double_side):
    """Same argument list as the reference's forward_soft_rasterize (cpp:59-91); numpy fp32
    arrays, outputs written in place.  `dist_eps` is already log(1/eps - 1)."""
    p = _params(faces, textures, image_size, near, far, eps, sigma_val, dist_eps, gamma_val,
                func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side)
    lib().sr_oracle_forward(_ptr(faces), _ptr(textures), _ptr(faces_info), _ptr(aggrs_info),
                            _ptr(soft_colors), ctypes.byref(p))
    return faces_info, aggrs_info, soft_colors


def backward(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
             grad_soft_colors, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
             gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side):
    """Same argument list as the reference's backward_soft_rasterize (cpp:94-132)."""
    p = _params(faces, textures, image_size, near, far, eps, sigma_val, dist_eps, gamma_val,
                func_id_dist, func_id_rgb, func_id_alpha, texture_sample_type, double_side)
    lib().sr_oracle_backward(_ptr(faces), _ptr(textures), _ptr(soft_colors), _ptr(faces_info),
                             _ptr(aggrs_info), _ptr(grad_faces), _ptr(grad_textures),
                             _ptr(grad_soft_colors), ctypes.byref(p))
    return grad_faces, grad_textures


def count_pairs(faces, image_size, sigma_val, dist_eps):
    tex = np.zeros((faces.shape[0], faces.shape[1], 1, 3), np.float32)
    p = _params(faces, tex, image_size, 1., 100., 1e-3, sigma_val, dist_eps, 1e-4, 2, 0, 2, 0, True)
    return int(lib().sr_oracle_count_pairs(_ptr(faces), ctypes.byref(p)))


def render(face_vertices, face_textures, image_size=256, background_color=(0, 0, 0), near=1.,
           far=100., fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func="euclidean",
           dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", aggr_func_alpha="prod",
           texture_type="surface", grad_soft_colors=None):
    """High-level helper following SoftRasterizeFunction.forward/backward
    (/root/reference/third-party/softras/soft_renderer/functional/soft_rasterize.py:12-102):
    allocates the buffers the way the reference does and returns a dict of numpy arrays."""
    fv = np.ascontiguousarray(face_vertices, np.float32).reshape(face_vertices.shape[0], -1, 9)
    B, F = fv.shape[:2]
    tex = np.ascontiguousarray(face_textures, np.float32).reshape(B, F, -1, 3)
    info = np.zeros((B, F, 27), np.float32)
    aggr = np.zeros((B, 2, image_size, image_size), np.float32)
    col = np.ones((B, 4, image_size, image_size), np.float32)
    for k in range(3):
        col[:, k] *= background_color[k]
    scal = (image_size, float(near), float(far), float(eps), float(sigma_val), DIST[dist_func],
            float(np.log(1. / dist_eps - 1.)), float(gamma_val), RGB[aggr_func_rgb],
            ALPHA[aggr_func_alpha], SAMPLE[texture_type], bool(fill_back))
    forward(fv, tex, info, aggr, col, *scal)
    out = {"soft_colors": col, "aggrs_info": aggr, "faces_info": info}
    if grad_soft_colors is not None:
        g = np.ascontiguousarray(grad_soft_colors, np.float32)
        gf = np.zeros_like(fv)
        gt = np.zeros_like(tex)
        backward(fv, tex, col, info, aggr, gf, gt, g, *scal)
        out["grad_faces"] = gf
        out["grad_textures"] = gt
    return out
