"""scp_amd/capi.py -- ctypes binding of libscp_hip.so (C ABI: include/scp_hip.h).

There is deliberately no fallback: if the library is missing or does not export a declared symbol
the import of this module's `lib()` raises.  Tensors are passed as raw device pointers and the
launch goes to torch's current HIP stream.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# SCP_HIP_LIB: an alternative build of the same library (A/B of kernel variants from tools/); the default is the in-tree build
LIB_PATH = os.environ.get("SCP_HIP_LIB") or os.path.join(_PKG, "lib", "libscp_hip.so")
ABI_VERSION = 8
CONV_PLANES_TILE = 32     # include/scp_hip.h: SCP_CONV_PLANES_TILE


class RasterParams(ctypes.Structure):
    """struct scp_raster_params"""
    _fields_ = [
        ("batch_size", ctypes.c_int), ("num_faces", ctypes.c_int), ("image_size", ctypes.c_int),
        ("texture_size", ctypes.c_int),
        ("near_", ctypes.c_float), ("far_", ctypes.c_float), ("eps", ctypes.c_float),
        ("sigma_val", ctypes.c_float), ("func_id_dist", ctypes.c_int), ("dist_eps", ctypes.c_float),
        ("gamma_val", ctypes.c_float), ("func_id_rgb", ctypes.c_int), ("func_id_alpha", ctypes.c_int),
        ("texture_sample_type", ctypes.c_int), ("double_side", ctypes.c_int),
    ]


ADAMW_MAX_CLASSES = 32     # include/scp_hip.h: SCP_ADAMW_MAX_CLASSES


class AdamWTensor(ctypes.Structure):
    """struct scp_adamw_tensor"""
    _fields_ = [("param", ctypes.c_ulonglong), ("flat_offset", ctypes.c_longlong), ("numel", ctypes.c_longlong),
                ("cls", ctypes.c_int), ("pad_", ctypes.c_int)]


class AdamWStep(ctypes.Structure):
    """struct scp_adamw_step"""
    _fields_ = [("lr_wd", ctypes.c_float * ADAMW_MAX_CLASSES), ("step_size", ctypes.c_float * ADAMW_MAX_CLASSES),
                ("inv_bias_correction2_sqrt", ctypes.c_float * ADAMW_MAX_CLASSES)]


class ConvPlanesDesc(ctypes.Structure):
    """struct scp_conv_planes_desc"""
    _fields_ = [("w", ctypes.c_ulonglong), ("planes_fwd", ctypes.c_ulonglong), ("planes_dgrad", ctypes.c_ulonglong),
                ("s_co", ctypes.c_longlong), ("s_ci", ctypes.c_longlong), ("s_ky", ctypes.c_longlong), ("s_kx", ctypes.c_longlong),
                ("block0", ctypes.c_longlong), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int), ("ksize", ctypes.c_int), ("pad_", ctypes.c_int)]


class CropDesc(ctypes.Structure):
    """struct scp_crop_desc"""
    _fields_ = [("img_off", ctypes.c_ulonglong), ("mask_off", ctypes.c_ulonglong), ("depth_off", ctypes.c_ulonglong),
                ("in_h", ctypes.c_int), ("in_w", ctypes.c_int), ("pad_top", ctypes.c_int), ("pad_left", ctypes.c_int),
                ("virt_h", ctypes.c_int), ("virt_w", ctypes.c_int)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_RP = ctypes.POINTER(RasterParams)

# name -> (restype, argtypes); must list every symbol include/scp_hip.h declares
SYMBOLS = {
    "scp_abi_version": (ctypes.c_int, []),
    "scp_last_error": (ctypes.c_char_p, []),
    "scp_stream_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "scp_stream_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "scp_soft_rasterize_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _RP, _P]),
    "scp_soft_rasterize_backward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _RP, _P]),
    "scp_soft_rasterize_count_pairs": (ctypes.c_int, [_P, _P, _RP, _P]),
    "scp_soft_rasterize_forward_dual": (ctypes.c_int, [_P] * 8 + [_RP, _P]),
    "scp_soft_rasterize_forward_f64": (ctypes.c_int, [_P] * 5 + [_RP, _P]),
    "scp_soft_rasterize_backward_f64": (ctypes.c_int, [_P] * 8 + [_RP, _P]),
    "scp_selftest_exact_division": (ctypes.c_int, [ctypes.c_ulonglong, ctypes.c_uint, _P, _P]),
    "scp_project_vertices_forward": (ctypes.c_int, [_P] * 5 + [_I] * 4 + [_P] * 3),
    "scp_project_vertices_backward": (ctypes.c_int, [_P] * 5 + [_I] * 4 + [_P] * 4),
    "scp_gradclip_workspace": (ctypes.c_size_t, []),
    "scp_gradclip": (ctypes.c_int, [_P, ctypes.c_longlong, _F, _P, _P, _P, _I, _F, _F, _F, _P, ctypes.c_size_t, _P, _P]),
    "scp_adamw_flat": (ctypes.c_int, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _F, _P]),
    "scp_conv_weight_planes_batch": (ctypes.c_int, [_P, _I, ctypes.c_longlong, _P]),
    "scp_selftest_mfma_load": (ctypes.c_int, [_I, _P, _I, _I, _P, _P]),
    "scp_selftest_packed_fp32": (ctypes.c_int, [_I, _P, _I, _I, _P]),
    "scp_vit_linear": (ctypes.c_int, [_P] * 7 + [ctypes.c_int] * 4 + [_P]),
    "scp_vit_linear_rows": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "scp_split_bf16x3_tiled_elements": (ctypes.c_size_t, [_I, _I]),
    "scp_split_bf16x3_tiled": (ctypes.c_int, [_P, _P, _I, _I, _P]),
    "scp_vit_linear_planes": (ctypes.c_int, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I, _I, _I, _P]),
    "scp_fvm_workspace": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "scp_fvm_forward": (ctypes.c_int, [_P] * 5 + [ctypes.c_float] * 2 + [ctypes.c_int] * 5 + [_P] * 9 + [ctypes.c_size_t, _P]),
    "scp_fvm_backward": (ctypes.c_int, [_P] * 5 + [ctypes.c_float] * 2 + [ctypes.c_int] * 5 + [_P] * 10),
    "scp_pp_softargmax_forward": (ctypes.c_int, [_P] * 5 + [_I, _F] + [_I] * 4 + [_P] * 3),
    "scp_pp_softargmax_backward": (ctypes.c_int, [_P] * 5 + [_I, _F] + [_I] * 4 + [_P] * 6),
    "scp_kernel_clock_begin": (ctypes.c_int, [_P, _I]),
    "scp_kernel_clock_end": (ctypes.c_int, []),
    "scp_split_bf16x3": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P]),
    "scp_row_mean_rstd": (ctypes.c_int, [_P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, _P]),
    "scp_softargmax_cols_workspace": (ctypes.c_size_t, [_I, _I, _I]),
    "scp_softargmax_cols_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _P, _P, _P, ctypes.c_size_t, _P]),
    "scp_softmax_rows_weighted_forward": (ctypes.c_int, [_P, _P, _I, _F, _I, _I, _I, _P, _P, _P]),
    "scp_nearest_point_workspace": (ctypes.c_size_t, [_I, _I]),
    "scp_nearest_point": (ctypes.c_int, [_P, _P, _I, _I, _I, _P, _P, ctypes.c_size_t, _P]),
    "scp_color_jitter_workspace": (ctypes.c_size_t, [_I]),
    "scp_color_jitter_normalize": (ctypes.c_int, [_P, _I, _I, _I, _P, _P, _P, _F, _P, _P, _I, _P, _P, ctypes.c_size_t, _P]),
    "scp_batchnorm_workspace": (ctypes.c_size_t, [ctypes.c_long, _I]),
    "scp_batchnorm_act_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _F, ctypes.c_long, _I, _I, _I, _P, _P, _P,
                                                _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "scp_batchnorm_act_backward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _I, _P, _P, _P, _P,
                                                 _P, ctypes.c_size_t, _P, _P]),
    "scp_batchnorm_act_forward_bf16": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _F, _F, ctypes.c_long, _I, _I, _I, _P, _P, _P,
                                                     _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "scp_batchnorm_act_backward_bf16": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_long, _I, _I, _I, _I, _P, _P, _P, _P,
                                                      _P, ctypes.c_size_t, _P, _P]),
    "scp_maxpool3x3s2_forward": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "scp_maxpool3x3s2_forward_bf16": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "scp_maxpool3x3s2_backward": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "scp_maxpool3x3s2_backward_bf16": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "scp_bias_leaky_relu_forward": (ctypes.c_int, [_P, _P, _F, ctypes.c_long, _I, _P]),
    "scp_bias_leaky_relu_forward_bf16": (ctypes.c_int, [_P, _P, _F, ctypes.c_long, _I, _P]),
    "scp_bias_leaky_relu_backward": (ctypes.c_int, [_P, _P, _F, ctypes.c_long, _I, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "scp_bias_leaky_relu_backward_bf16": (ctypes.c_int, [_P, _P, _F, ctypes.c_long, _I, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "scp_conv_nhwc_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, ctypes.c_size_t, _P]),
    "scp_conv_nhwc_splitk_workspace": (ctypes.c_size_t, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "scp_conv_nhwc_forward_bn": (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P,
                                                ctypes.c_size_t, _P, _P, ctypes.c_size_t, _P]),
    "scp_batchnorm_apply": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_long, _I, _I, _P, _P]),
    "scp_conv_weight_planes": (ctypes.c_int, [_P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _I, _I, _I,
                                              _P, _P, _P]),
    "scp_conv_nhwc_dgrad_stride2": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "scp_stem_conv_tiles": (ctypes.c_int, [_I, _I, _I]),
    "scp_stem_conv_forward_bn": (ctypes.c_int, [_P, _P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, _I, _I, _I, _P, _P, _P, _P, _P, ctypes.c_float,
                                                ctypes.c_float, _P, _P, _P, _P, _P, ctypes.c_size_t, _P, _P]),
    "scp_stem_conv_weight_grad_workspace": (ctypes.c_size_t, [_I, _I, _I]),
    "scp_stem_conv_weight_grad": (ctypes.c_int, [_P, _P, _P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, ctypes.c_size_t, _I, _I, _I, _P]),
    "scp_conv1x1_nhwc_dgrad_stride2": (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "scp_conv_nhwc_partial_rows": (ctypes.c_int, [_I, _I, _I, _I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "scp_conv_nhwc_weight_grad_workspace": (ctypes.c_size_t, [_I, _I, _I, _I, _I, _I, _I]),
    "scp_conv_nhwc_weight_grad": (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_size_t, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "scp_posefit_workspace": (ctypes.c_size_t, [_I, _I, _I]),
    "scp_ransac_hypotheses": (ctypes.c_int, [_P, _P, _I, _I, _P, _I, _P, _P]),
    "scp_ransac_score": (ctypes.c_int, [_P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "scp_umeyama_fit_inliers": (ctypes.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "scp_crop_resize_batch": (ctypes.c_int, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "scp_mutual_argmax_workspace": (ctypes.c_size_t, [_I, _I]),
    "scp_mutual_argmax": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P, ctypes.c_size_t, _P]),
    "scp_mutual_nn_fused_workspace": (ctypes.c_size_t, [_I, _I]),
    "scp_mutual_nn_fused": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, ctypes.c_size_t, _P]),
    "scp_upsample2x_bilinear_backward": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "scp_upsample2x_bilinear_forward": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "scp_upsample2x_bilinear_forward_bf16": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "scp_upsample2x_bilinear_backward_bf16": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "scp_vit_attention_bf16_forward": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "scp_add_layernorm_forward": (ctypes.c_int, [_P, _P, _P, _P, _F, ctypes.c_long, _I, _P, _P, _P]),
    "scp_vit_attention_forward": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "scp_vit_attention_forward_rows": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P]),
    "scp_image_losses_parts": (ctypes.c_int, []),
    "scp_image_losses_forward": (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "scp_image_losses_backward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "scp_image_losses_backward_scale": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "scp_texture_loss_forward": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "scp_texture_loss_backward": (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "scp_vit_attention_split_workspace": (ctypes.c_size_t, [_I, _I, _I]),
    "scp_vit_attention_split_forward_presplit": (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, ctypes.c_size_t, _P]),
    "scp_vit_linear_qkv": (ctypes.c_int, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _F, _I, _P]),
    "scp_vit_attention_split_forward": (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, ctypes.c_size_t, _P]),
    "scp_dual_softmax_backward": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _F,
                                                _I, _I, _I, _P]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "scp_amd: %s is missing -- build it with `python self-corr-pose_amd/build.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.scp_abi_version() != ABI_VERSION:
            raise RuntimeError("scp_amd: libscp_hip.so ABI %d != expected %d" % (handle.scp_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError("scp_amd: %s failed (hip error %d): %s" % (what, code, lib().scp_last_error().decode()))


def current_stream():
    """the stream the C ABI launches on: the CURRENT device's current stream.  The raw HIP launches inside libscp_hip.so
    go to the current device, so tensors must live there: Trainer / Tester call torch.cuda.set_device(their device), and
    dev_ptr() rejects a tensor from another device instead of faulting on a foreign pointer."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_TICKETS = {}
_TICKET_WORDS = 8192
# set by scp_amd.graphed while a HIP graph is being captured: launches recorded into a graph keep their ticket word for every
# replay, so they get words of their own that the round-robin pool never hands out again
CAPTURING = False
_GRAPH_TICKETS = {}
_GRAPH_TICKET_WORDS = 65536


def ticket(device):
    """pointer to one ZEROED 32-bit device word for a kernel that elects its last workgroup (include/scp_hip.h: `ticket`);
    the kernel leaves it zero.  A pool of 8192 words per device handed out round-robin: a word is reused only after 8191
    later launches have been enqueued."""
    import torch
    key = device.index if device.index is not None else torch.cuda.current_device()
    if CAPTURING:
        pool = _GRAPH_TICKETS.get(key)
        if pool is None:
            raise RuntimeError("scp_amd.capi: reserve_graph_tickets(device) must run before a capture (it allocates and synchronises)")
        i = pool[1]
        if i >= _GRAPH_TICKET_WORDS:
            raise RuntimeError("scp_amd.capi: graph ticket words exhausted")
        pool[1] = i + 1
        return ctypes.c_void_p(pool[0].data_ptr() + 4 * i)
    pool = _TICKETS.get(key)
    if pool is None:
        buf = torch.zeros(_TICKET_WORDS, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()      # the fill must be complete before ANY stream uses a word
        pool = _TICKETS[key] = [buf, 0]
    i = pool[1]
    pool[1] = (i + 1) % _TICKET_WORDS
    return ctypes.c_void_p(pool[0].data_ptr() + 4 * i)


def reserve_graph_tickets(device):
    """the zeroed ticket words of launches that get captured into HIP graphs (see CAPTURING); call before the first capture"""
    import torch
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _GRAPH_TICKETS:
        buf = torch.zeros(_GRAPH_TICKET_WORDS, dtype=torch.int32, device=device)
        torch.cuda.current_stream(device).synchronize()
        _GRAPH_TICKETS[key] = [buf, 0]


def dev_ptr(t, name):
    """raw device pointer of a contiguous fp32 CUDA tensor; the reference's CHECK_INPUT
    (soft_rasterize_cuda.cpp:54-56) raises RuntimeError for CPU / non-contiguous tensors, so do we"""
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (the gfx950 kernels are fp32-only)" % name)
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError("%s lives on cuda:%d but the current device is cuda:%d (torch.cuda.set_device first)"
                           % (name, t.device.index, torch.cuda.current_device()))
    return ctypes.c_void_p(t.data_ptr())


def dev_ptr64(t, name):
    """dev_ptr for the double entry points (scp_soft_rasterize_*_f64)"""
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != torch.float64:
        raise RuntimeError("%s must be float64 like the other tensors of this call" % name)
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError("%s lives on cuda:%d but the current device is cuda:%d" % (name, t.device.index, torch.cuda.current_device()))
    return ctypes.c_void_p(t.data_ptr())


def dev_ptr_any(t):
    """raw device pointer of a dense fp32 CUDA tensor in ANY dense layout (e.g. channels_last: the kernel's NHWC view)"""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("expected a float32 CUDA tensor")
    if not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("expected a dense (contiguous or channels_last) tensor")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError("tensor lives on cuda:%d but the current device is cuda:%d" % (t.device.index, torch.cuda.current_device()))
    return ctypes.c_void_p(t.data_ptr())


def opt_ptr(t, name):
    """like dev_ptr, but None -> NULL"""
    return ctypes.c_void_p(0) if t is None else dev_ptr(t, name)
