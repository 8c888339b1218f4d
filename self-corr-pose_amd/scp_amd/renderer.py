"""scp_amd/renderer.py -- the four SoftRas passes of one training step.

API and semantics of model/module/renderer.py (Renderer.__init__ :11-27, render_mean_mesh :29-36,
render_all :38-73).  The pass configurations (sigma, gamma, aggregation, background) are the
reference's.  Exact savings taken here: the canonical-xyz ("hardtex") pass runs without autograd
because its backward is identically zero (detached vertices/colours, hard colours carry no
geometry gradient and its alpha is only thresholded) -- SURVEY F8.
"""
import torch
import torch.nn.functional as F

from . import soft_renderer as sr
from .losses import pinhole_cam, render


class Renderer:
    def __init__(self, opts, mesh):
        self.opts = opts
        common = dict(image_size=opts.img_size, camera_mode="look_at", perspective=False, light_mode="vertex",
                      light_intensity_ambient=1., light_intensity_directionals=0.)
        self.renderer_mask = sr.SoftRenderer(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="hard", **common)
        self.renderer_depth = sr.SoftRenderer(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", **common)
        self.renderer_softtex = sr.SoftRenderer(sigma_val=1e-3, gamma_val=1e-2, aggr_func_rgb="softmax", **common)
        self.renderer_hardtex = sr.SoftRenderer(sigma_val=1e-4, gamma_val=1e-3, aggr_func_rgb="hard", **common)
        self.renderer_depth.rasterizer.background_color = [1, 1, 1]
        self.renderer_softtex.rasterizer.background_color = [1, 1, 1]
        self.mesh = mesh

    def render_mean_mesh(self, foc_crop, pp_crop, rotation, translation):
        bsz = rotation.shape[0]
        mean_v = self.mesh.mean_v[None].expand(bsz, -1, -1)
        faces = self.mesh.faces[None].expand(bsz, -1, -1)
        return render(self.renderer_depth, mean_v, faces, None, foc_crop, pp_crop, rotation, translation,
                      rotation_detach=True, translation_detach=True, render_depth=True)

    def render_all(self, pred_v, faces, tex, foc_crop, pp_crop, rotation, translation, scale):
        cam = (foc_crop, pp_crop, rotation, translation)
        # The mask pass and the depth pass share sigma, distance function and alpha aggregation, so
        # their alpha planes are bit-identical (checked in tests/test_softras_gpu.py) and the
        # mask pass' gradient reaches the geometry only through alpha: taking the mask from the depth
        # pass' alpha channel is the same forward value and -- by linearity of the backward in the
        # incoming alpha gradient -- the same gradient, with one rasterisation + one backward less.
        fuse_mask = bool(self.opts.use_depth) and getattr(self, "share_mask_with_depth", True)
        if not fuse_mask:
            mask_render = render(self.renderer_mask, pred_v, faces, None, *cam, render_mask=True)[:, -1]

        if tex is not None:
            tex_out = render(self.renderer_softtex, pred_v, faces, tex, *cam, texture_type=self.mesh.texture_type)
            tex_mask, tex_render = tex_out[:, -1], tex_out[:, :3]
        else:
            tex_mask = tex_render = None

        depth_out = render(self.renderer_depth, pred_v, faces, None, *cam, render_depth=True, texture_type="vertex")
        if fuse_mask:
            mask_render = depth_out[:, 3]
        if not self.opts.use_depth:
            depth_out = depth_out.detach()
        depth_mask, depth_render = depth_out[:, 3], depth_out[:, 2].clone()

        with torch.no_grad():  # zero backward in the reference as well (F8)
            canon = pred_v.detach()
            match_out = render(self.renderer_hardtex, canon, faces, canon, *cam, texture_type="vertex")
        match_mask, match_gt = match_out[:, -1], match_out[:, :3]

        # projected vertex positions: differentiable w.r.t. rotation / translation (renderer.py:63-67)
        cam_v = canon.bmm(rotation) + translation
        imatch_gt = pinhole_cam(cam_v, pp_crop, foc_crop)[:, :, :2].permute(0, 2, 1)   # b,2,n
        with torch.no_grad():  # visibility weight, detached in the reference (:69-71)
            seen = F.grid_sample(depth_render[:, None], imatch_gt.permute(0, 2, 1)[:, None], align_corners=False)[:, 0, 0]
            depth_weight = (-5 * F.relu(cam_v[:, :, 2] - seen)).exp()

        return mask_render, tex_render, depth_render, match_gt, imatch_gt, tex_mask, depth_mask, match_mask, depth_weight
