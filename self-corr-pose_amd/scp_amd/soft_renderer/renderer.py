"""scp_amd.soft_renderer.renderer -- SoftRenderer and its three stages.

Behaviour of (reference paths under third-party/softras/soft_renderer/):
  renderer.py:47-101     SoftRenderer (lighting -> transform -> rasteriser), render_mesh
  lighting.py:34-66      Lighting (ambient + one directional light; 'surface' or 'vertex')
  transform.py:29-48,85-119  LookAt / Transform ('look_at' camera; 'look' and 'projection' are
                         not used by the trainer and raise)
  rasterizer.py:10-54    SoftRasterizer
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as srf
from .mesh import Mesh


class Lighting(nn.Module):
    def __init__(self, light_mode="surface", intensity_ambient=0.5, color_ambient=(1, 1, 1),
                 intensity_directionals=0.5, color_directionals=(1, 1, 1), directions=(0, 1, 0)):
        super().__init__()
        if light_mode not in ("surface", "vertex"):
            raise ValueError("Lighting mode only support surface and vertex")
        self.light_mode = light_mode
        self.intensity_ambient, self.color_ambient = intensity_ambient, color_ambient
        self.intensity_directionals, self.color_directionals = intensity_directionals, color_directionals
        self.directions = directions

    def _light(self, like, normals_fn):
        light = srf.ambient_lighting(torch.zeros_like(like, dtype=torch.float32),
                                     self.intensity_ambient, self.color_ambient)
        if self.intensity_directionals != 0:
            # the reference evaluates the normals even for a zero-intensity light and multiplies
            # by 0 (lighting.py:59-63); skipping that is exact for finite normals
            light = srf.directional_lighting(light, normals_fn(), self.intensity_directionals,
                                             self.color_directionals, self.directions)
        return light

    def _is_identity(self):
        """ambient 1.0 x white and no directional light (every renderer of model/module/renderer.py:13-26): the light is exactly 1.0
        per channel and textures * 1.0 are the textures -- nothing to launch"""
        try:
            return (float(self.intensity_directionals) == 0.0 and float(self.intensity_ambient) == 1.0
                    and all(float(c) == 1.0 for c in self.color_ambient))
        except (TypeError, ValueError):
            return False

    def forward(self, mesh):
        if self._is_identity() and mesh.textures is not None and mesh.textures.dtype == torch.float32:
            return mesh
        if self.light_mode == "surface":
            light = self._light(mesh.faces, lambda: mesh.surface_normals)
            mesh.textures = mesh.textures * light[:, :, None, :]
        else:
            light = self._light(mesh.vertices, lambda: mesh.vertex_normals)
            mesh.textures = mesh.textures * light
        return mesh


class LookAt(nn.Module):
    def __init__(self, perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None):
        super().__init__()
        self.perspective, self.viewing_angle, self.viewing_scale = perspective, viewing_angle, viewing_scale
        self._eye = eye if eye is not None else [0, 0, -(1. / math.tan(math.radians(viewing_angle)) + 1)]

    def _pure_translation(self):
        """The trainer's camera (renderer.py:13-26: look_at, orthographic, default eye on the -z axis looking at the origin with +y up)
        is a pure translation: the look-at rotation is the identity EXACTLY (normalize((0, 0, e)) = (0, 0, 1) in IEEE arithmetic:
        sqrt(RN(e^2)) = |e| whether the square is kept in fp32 or fp64) and the orthographic scale is 1, so
        orthogonal(look_at(v)) = v - eye bit for bit (x * 1 + y * 0 + z * 0 adds exact zeros).  One launch instead of ~14
        (normalize x3, cross x2, stack, matmul, three selects, two multiplies, stack) and one backward node instead of ~10."""
        eye = self._eye
        if self.perspective or self.viewing_scale != 1.0 or not isinstance(eye, (list, tuple)) or len(eye) != 3:
            return False
        try:
            ex, ey, ez = (float(t) for t in eye)
        except (TypeError, ValueError):
            return False
        import numpy as np
        e32 = np.float32(ez)
        return ex == 0.0 and ey == 0.0 and ez < 0.0 and np.float32(-e32) / np.sqrt(e32 * e32, dtype=np.float32) == np.float32(1.0)

    def forward(self, vertices):
        if self._pure_translation():
            return vertices - srf.const_tensor(self._eye, torch.float32, vertices.device).reshape(1, 1, 3)
        vertices = srf.look_at(vertices, self._eye)
        if self.perspective:
            return srf.perspective(vertices, angle=self.viewing_angle)
        return srf.orthogonal(vertices, scale=self.viewing_scale)


class Transform(nn.Module):
    def __init__(self, camera_mode="projection", P=None, dist_coeffs=None, orig_size=512,
                 perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None,
                 camera_direction=(0, 0, 1)):
        super().__init__()
        self.camera_mode = camera_mode
        if camera_mode == "look_at":
            self.transformer = LookAt(perspective, viewing_angle, viewing_scale, eye)
        elif camera_mode in ("projection", "look"):
            raise NotImplementedError("camera_mode=%r is not on the training hot path" % camera_mode)
        else:
            raise ValueError("Camera mode has to be one of projection, look or look_at")

    def forward(self, mesh):
        mesh.vertices = self.transformer(mesh.vertices)
        return mesh

    def set_eyes(self, eyes):
        self.transformer._eye = eyes


class SoftRasterizer(nn.Module):
    def __init__(self, image_size=256, background_color=[0, 0, 0], near=1, far=100,
                 anti_aliasing=False, fill_back=False, eps=1e-3, sigma_val=1e-5,
                 dist_func="euclidean", dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax",
                 aggr_func_alpha="prod", texture_type="surface"):
        super().__init__()
        if dist_func not in srf.DIST_IDS:
            raise ValueError("Distance function only support hard, euclidean and barycentric")
        if aggr_func_rgb not in srf.RGB_IDS:
            raise ValueError("Aggregate function(rgb) only support hard and softmax")
        if aggr_func_alpha not in srf.ALPHA_IDS:
            raise ValueError("Aggregate function(a) only support hard, prod and sum")
        if texture_type not in srf.SAMPLE_IDS:
            raise ValueError("Texture type only support surface and vertex")
        self.image_size, self.background_color = image_size, background_color
        self.near, self.far, self.anti_aliasing, self.fill_back, self.eps = near, far, anti_aliasing, fill_back, eps
        self.sigma_val, self.dist_func, self.dist_eps = sigma_val, dist_func, dist_eps
        self.gamma_val, self.aggr_func_rgb, self.aggr_func_alpha = gamma_val, aggr_func_rgb, aggr_func_alpha
        self.texture_type = texture_type

    def forward(self, mesh, mode=None):
        size = self.image_size * (2 if self.anti_aliasing else 1)
        images = srf.soft_rasterize(mesh.face_vertices, mesh.face_textures, size, self.background_color,
                                    self.near, self.far, self.fill_back, self.eps, self.sigma_val,
                                    self.dist_func, self.dist_eps, self.gamma_val, self.aggr_func_rgb,
                                    self.aggr_func_alpha, self.texture_type)
        if self.anti_aliasing:
            images = F.avg_pool2d(images, kernel_size=2, stride=2)
        return images


class SoftRenderer(nn.Module):
    def __init__(self, image_size=256, background_color=[0, 0, 0], near=1, far=100,
                 anti_aliasing=False, fill_back=True, eps=1e-3, sigma_val=1e-5, dist_func="euclidean",
                 dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", aggr_func_alpha="prod",
                 texture_type="surface", camera_mode="projection", P=None, dist_coeffs=None,
                 orig_size=512, perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None,
                 camera_direction=[0, 0, 1], light_mode="surface", light_intensity_ambient=0.5,
                 light_color_ambient=[1, 1, 1], light_intensity_directionals=0.5,
                 light_color_directionals=[1, 1, 1], light_directions=[0, 1, 0]):
        super().__init__()
        self.lighting = Lighting(light_mode, light_intensity_ambient, light_color_ambient,
                                 light_intensity_directionals, light_color_directionals, light_directions)
        self.transform = Transform(camera_mode, P, dist_coeffs, orig_size, perspective, viewing_angle,
                                   viewing_scale, eye, camera_direction)
        self.rasterizer = SoftRasterizer(image_size, background_color, near, far, anti_aliasing,
                                         fill_back, eps, sigma_val, dist_func, dist_eps, gamma_val,
                                         aggr_func_rgb, aggr_func_alpha, texture_type)

    def set_sigma(self, sigma):
        self.rasterizer.sigma_val = sigma

    def set_gamma(self, gamma):
        self.rasterizer.gamma_val = gamma

    def set_texture_mode(self, mode):
        assert mode in ("vertex", "surface"), "Mode only support surface and vertex"
        self.lighting.light_mode = mode
        self.rasterizer.texture_type = mode

    def render_mesh(self, mesh, mode=None):
        self.set_texture_mode(mesh.texture_type)
        mesh = self.lighting(mesh)
        mesh = self.transform(mesh)
        return self.rasterizer(mesh, mode)

    def render_mesh_with_hard(self, mesh, hard_renderer, hard_textures):
        """render_mesh(mesh) with this (softmax-rgb, vertex-texture) renderer AND hard_renderer.render_mesh(Mesh(same
        vertices, same faces, hard_textures, 'vertex')) in one rasteriser launch.  Legal only when both renderers
        produce the same coverage: same image size, sigma, distance / alpha functions, clipping and camera -- checked."""
        ra, rb = self.rasterizer, hard_renderer.rasterizer
        same = ("image_size", "near", "far", "anti_aliasing", "fill_back", "eps", "sigma_val", "dist_func", "dist_eps",
                "aggr_func_alpha")
        if (any(getattr(ra, k) != getattr(rb, k) for k in same) or ra.aggr_func_rgb != "softmax" or rb.aggr_func_rgb != "hard"
                or mesh.texture_type != "vertex" or ra.anti_aliasing
                or self.transform.transformer._eye != hard_renderer.transform.transformer._eye):
            raise ValueError("render_mesh_with_hard: the two renderers do not share their coverage")
        self.set_texture_mode("vertex")
        hard_renderer.set_texture_mode("vertex")
        hard = hard_renderer.lighting(Mesh(mesh.vertices, mesh.faces, hard_textures, texture_type="vertex"))
        mesh = self.transform(self.lighting(mesh))
        return srf.soft_rasterize_dual(mesh.face_vertices, mesh.face_textures, srf.face_vertices(hard.textures, hard.faces),
                                       ra.image_size, ra.background_color, rb.background_color, ra.near, ra.far,
                                       ra.fill_back, ra.eps, ra.sigma_val, ra.dist_func, ra.dist_eps, ra.gamma_val,
                                       ra.aggr_func_alpha)

    def forward(self, vertices, faces, textures=None, mode=None, texture_type="surface"):
        return self.render_mesh(Mesh(vertices, faces, textures=textures, texture_type=texture_type), mode)
