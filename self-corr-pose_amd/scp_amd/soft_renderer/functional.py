"""scp_amd.soft_renderer.functional -- the `soft_renderer.functional` (srf) surface the hot path uses.

Own implementation of the behaviour of (reference paths under third-party/softras/soft_renderer/):
  functional/soft_rasterize.py:9-119   SoftRasterizeFunction / soft_rasterize
  functional/face_vertices.py:4-22     face_vertices
  functional/vertex_normals.py:4-36    vertex_normals
  functional/look_at.py:6-62           look_at
  functional/orthogonal.py:4-18        orthogonal
  functional/perspective.py            perspective
  functional/ambient_lighting.py, directional_lighting.py
Texture atlas IO / voxelisation (load_obj, save_obj with textures, voxelization) are out of scope:
the trainer never calls them.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .cuda import soft_rasterize as _native
from ..hostconst import const_tensor

DIST_IDS = {"hard": 0, "barycentric": 1, "euclidean": 2}
RGB_IDS = {"hard": 0, "softmax": 1}
ALPHA_IDS = {"hard": 0, "sum": 1, "prod": 2}
SAMPLE_IDS = {"surface": 0, "vertex": 1}


class SoftRasterizeFunction(Function):
    """Autograd node around the native rasteriser; buffer protocol as in the reference
    (caller allocates: soft_colors = background RGB + alpha 1, infos/grads = 0), but every
    buffer is created directly on the device (the reference builds them on the host and copies,
    soft_rasterize.py:47-53)."""

    @staticmethod
    def forward(ctx, face_vertices, textures, image_size=256, background_color=[0, 0, 0], near=1,
                far=100, fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func="euclidean",
                dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", aggr_func_alpha="prod",
                texture_type="surface"):
        ctx.scalars = (int(image_size), float(near), float(far), float(eps), float(sigma_val),
                       DIST_IDS[dist_func], float(np.log(1. / dist_eps - 1.)), float(gamma_val),
                       RGB_IDS[aggr_func_rgb], ALPHA_IDS[aggr_func_alpha], SAMPLE_IDS[texture_type],
                       bool(fill_back))
        nb, nf = face_vertices.shape[:2]
        fv = face_vertices.detach().reshape(nb, nf, 9).contiguous()
        tex = textures.detach().reshape(nb, nf, -1, 3).contiguous().to(fv.dtype)
        dev, dt = fv.device, fv.dtype           # float or double, like the reference (AT_DISPATCH_FLOATING_TYPES)
        faces_info = torch.zeros(nb, nf, 27, dtype=dt, device=dev)
        aggrs_info = torch.zeros(nb, 2, image_size, image_size, dtype=dt, device=dev)
        soft_colors = torch.ones(nb, 4, image_size, image_size, dtype=dt, device=dev)
        for k in range(3):
            if background_color[k] != 1:
                soft_colors[:, k].fill_(float(background_color[k]))
        _native.forward_soft_rasterize(fv, tex, faces_info, aggrs_info, soft_colors, *ctx.scalars)
        ctx.save_for_backward(fv, tex, soft_colors, faces_info, aggrs_info)
        ctx.in_shapes = (face_vertices.shape, textures.shape)
        return soft_colors

    @staticmethod
    def backward(ctx, grad_soft_colors):
        fv, tex, soft_colors, faces_info, aggrs_info = ctx.saved_tensors
        grad_faces = torch.zeros_like(fv)
        grad_textures = torch.zeros_like(tex)
        _native.backward_soft_rasterize(fv, tex, soft_colors, faces_info, aggrs_info, grad_faces,
                                        grad_textures, grad_soft_colors.contiguous(), *ctx.scalars)
        return (grad_faces.reshape(ctx.in_shapes[0]), grad_textures.reshape(ctx.in_shapes[1])) + (None,) * 13


def soft_rasterize(face_vertices, textures, image_size=256, background_color=[0, 0, 0], near=1,
                   far=100, fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func="euclidean",
                   dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", aggr_func_alpha="prod",
                   texture_type="surface"):
    return SoftRasterizeFunction.apply(face_vertices, textures, image_size, background_color, near,
                                       far, fill_back, eps, sigma_val, dist_func, dist_eps, gamma_val,
                                       aggr_func_rgb, aggr_func_alpha, texture_type)


class SoftRasterizeDualFunction(Function):
    """SoftRasterizeFunction for the depth-style pass (softmax rgb, vertex textures) that ALSO emits the hard-rgb image of
    a second vertex-texture set over the same coverage (model/module/renderer.py:52-61: `renderer_depth` and
    `renderer_hardtex` share sigma, distance and alpha functions and the projected geometry) -- one launch
    instead of two.  The second image carries no gradient (its inputs are detached in the reference and hard colours have
    none w.r.t. geometry); backward is the primary pass' backward, unchanged."""

    @staticmethod
    def forward(ctx, face_vertices, textures, textures_hard, image_size, background_color, background_color_hard,
                near, far, fill_back, eps, sigma_val, dist_func, dist_eps, gamma_val, aggr_func_alpha):
        ctx.scalars = (int(image_size), float(near), float(far), float(eps), float(sigma_val),
                       DIST_IDS[dist_func], float(np.log(1. / dist_eps - 1.)), float(gamma_val),
                       RGB_IDS["softmax"], ALPHA_IDS[aggr_func_alpha], SAMPLE_IDS["vertex"], bool(fill_back))
        nb, nf = face_vertices.shape[:2]
        fv = face_vertices.detach().reshape(nb, nf, 9).contiguous()
        tex = textures.detach().reshape(nb, nf, 3, 3).contiguous()
        tex_h = textures_hard.detach().reshape(nb, nf, 3, 3).contiguous()
        dev = fv.device
        faces_info = torch.zeros(nb, nf, 27, dtype=torch.float32, device=dev)
        aggrs = torch.zeros(2, nb, 2, image_size, image_size, dtype=torch.float32, device=dev)
        cols = torch.ones(2, nb, 4, image_size, image_size, dtype=torch.float32, device=dev)
        for i, bg in enumerate((background_color, background_color_hard)):
            for k in range(3):
                if bg[k] != 1:
                    cols[i, :, k].fill_(float(bg[k]))
        soft, hard = cols[0], cols[1]
        _native.forward_soft_rasterize_dual(fv, tex, faces_info, aggrs[0], soft, tex_h, aggrs[1], hard, *ctx.scalars)
        ctx.save_for_backward(fv, tex, soft, faces_info, aggrs[0])
        ctx.in_shapes = (face_vertices.shape, textures.shape)
        ctx.mark_non_differentiable(hard)
        return soft, hard

    @staticmethod
    def backward(ctx, grad_soft_colors, _grad_hard):
        fv, tex, soft_colors, faces_info, aggrs_info = ctx.saved_tensors
        grad_faces = torch.zeros_like(fv)
        grad_textures = torch.zeros_like(tex)
        _native.backward_soft_rasterize(fv, tex, soft_colors, faces_info, aggrs_info, grad_faces,
                                        grad_textures, grad_soft_colors.contiguous(), *ctx.scalars)
        return (grad_faces.reshape(ctx.in_shapes[0]), grad_textures.reshape(ctx.in_shapes[1])) + (None,) * 13


def soft_rasterize_dual(face_vertices, textures, textures_hard, image_size=256, background_color=[0, 0, 0],
                        background_color_hard=[0, 0, 0], near=1, far=100, fill_back=True, eps=1e-3, sigma_val=1e-5,
                        dist_func="euclidean", dist_eps=1e-4, gamma_val=1e-4, aggr_func_alpha="prod"):
    """-> (softmax-rgb image of `textures`, hard-rgb image of `textures_hard`), both [B,4,S,S]"""
    return SoftRasterizeDualFunction.apply(face_vertices, textures, textures_hard, image_size, background_color,
                                           background_color_hard, near, far, fill_back, eps, sigma_val, dist_func,
                                           dist_eps, gamma_val, aggr_func_alpha)


class _SharedTopologyGather(Function):
    """vertices[:, faces] for ONE face list shared by the batch.  Backward = incidence^T-matmul
    (a [V, 3F] 0/1 matrix times the face gradients): deterministic and ~30x faster on the GPU than the
    sorted index_put autograd emits for an advanced-index gather."""
    _incidence = {}

    @staticmethod
    def forward(ctx, vertices, faces1):
        ctx.save_for_backward(faces1)
        ctx.nv = vertices.shape[1]
        return vertices[:, faces1]

    @staticmethod
    def backward(ctx, grad):
        faces1, = ctx.saved_tensors
        key = (faces1.data_ptr(), faces1.shape[0], ctx.nv, str(grad.device), faces1._version)
        inc = _SharedTopologyGather._incidence.get(key)
        if inc is None:
            inc = torch.zeros(ctx.nv, faces1.numel(), dtype=grad.dtype, device=grad.device)
            inc[faces1.reshape(-1), torch.arange(faces1.numel(), device=grad.device)] = 1
            _SharedTopologyGather._incidence = {key: inc}
        nb = grad.shape[0]
        return torch.matmul(inc, grad.reshape(nb, -1, grad.shape[-1])), None


def face_vertices(vertices, faces):
    """[B,V,C], [B,F,3] -> [B,F,3,C]"""
    assert vertices.dim() == 3 and faces.dim() == 3 and vertices.shape[0] == faces.shape[0]
    assert vertices.shape[2] == 3 and faces.shape[2] == 3
    nb, nv = vertices.shape[:2]
    if faces.stride(0) == 0 and vertices.is_cuda:        # one topology expanded over the batch (the trainer's case)
        return _SharedTopologyGather.apply(vertices, faces[0].long())
    flat = faces.long() + (torch.arange(nb, device=vertices.device) * nv)[:, None, None]
    return vertices.reshape(nb * nv, 3)[flat]


def vertex_normals(vertices, faces):
    nb, nv = vertices.shape[:2]
    fv = face_vertices(vertices, faces).reshape(-1, 3, 3)
    flat = (faces.long() + (torch.arange(nb, device=vertices.device) * nv)[:, None, None]).reshape(-1, 3)
    normals = torch.zeros(nb * nv, 3, device=vertices.device, dtype=vertices.dtype)
    normals.index_add_(0, flat[:, 1], torch.cross(fv[:, 2] - fv[:, 1], fv[:, 0] - fv[:, 1], dim=1))
    normals.index_add_(0, flat[:, 2], torch.cross(fv[:, 0] - fv[:, 2], fv[:, 1] - fv[:, 2], dim=1))
    normals.index_add_(0, flat[:, 0], torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=1))
    return F.normalize(normals, eps=1e-6, dim=1).reshape(nb, nv, 3)


def _as_rows(x, nb, device):
    x = const_tensor(x, torch.float32, device)
    return x[None].repeat(nb, 1) if x.dim() == 1 else x


def look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0)):
    if vertices.dim() != 3:
        raise ValueError("vertices Tensor should have 3 dimensions")
    nb, dev = vertices.shape[0], vertices.device
    eye, at, up = _as_rows(eye, nb, dev), _as_rows(at, nb, dev), _as_rows(up, nb, dev)
    z_axis = F.normalize(at - eye, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    r = torch.stack((x_axis, y_axis, z_axis), dim=1)
    return torch.matmul(vertices - eye[:, None, :], r.transpose(1, 2))


def orthogonal(vertices, scale):
    if vertices.dim() != 3:
        raise ValueError("vertices Tensor should have 3 dimensions")
    return torch.stack((vertices[:, :, 0] * scale, vertices[:, :, 1] * scale, vertices[:, :, 2]), dim=2)


def perspective(vertices, angle=30.):
    if vertices.dim() != 3:
        raise ValueError("vertices Tensor should have 3 dimensions")
    width = math.tan(math.radians(angle))
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


def ambient_lighting(light, light_intensity=0.5, light_color=(1, 1, 1)):
    color = const_tensor(light_color, torch.float32, light.device).reshape(1, 1, 3)
    return light + light_intensity * color


def directional_lighting(light, normals, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
    color = const_tensor(light_color, torch.float32, light.device).reshape(1, 1, 3)
    direction = const_tensor(light_direction, torch.float32, light.device).reshape(1, 1, 3)
    cosine = F.relu(torch.sum(normals * direction, dim=2))
    return light + light_intensity * (color * cosine[:, :, None])
