"""scp_amd/renderer.py -- the four SoftRas passes of one training step.

API and semantics of model/module/renderer.py (Renderer.__init__ :11-27, render_mean_mesh :29-36,
render_all :38-73).  The pass configurations (sigma, gamma, aggregation, background) are the
reference's.  Exact savings taken here: the mask is the depth pass' alpha plane, the
canonical-xyz ("hardtex") pass is a second colour output of the depth pass' launch and carries no
backward (detached vertices/colours, hard colours have no geometry gradient, its alpha is only
thresholded) -- four reference passes = two forward launches + two backward launches.
"""
import torch
import torch.nn.functional as F

from . import soft_renderer as sr
from .losses import pinhole_cam, project_for_render, project_vertices, render


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

    def render_texture(self, pred_v, faces, tex, foc_crop, pp_crop, rotation, translation):
        """the soft-texture pass (sigma 1e-3, gamma 1e-2, softmax colours): (tex_render [B,3,H,W], tex_mask [B,H,W]).  It shares
        nothing with the depth group below but its inputs, so MeshNet runs it (and the texture loss) on a side stream."""
        if tex is None:
            return None, None
        tex_out = self.render_texture_raw(pred_v, faces, tex, foc_crop, pp_crop, rotation, translation)
        return tex_out[:, :3], tex_out[:, -1]

    def render_texture_raw(self, pred_v, faces, tex, foc_crop, pp_crop, rotation, translation):
        """the soft-texture pass' [B,4,H,W] output (rgb + alpha) as the rasteriser leaves it: what fused_losses.texture_loss reads"""
        return render(self.renderer_softtex, pred_v, faces, tex, foc_crop, pp_crop, rotation, translation,
                      texture_type=self.mesh.texture_type)

    def shares_mask_with_depth(self):
        return bool(self.opts.use_depth) and getattr(self, "share_mask_with_depth", True)

    def render_depth_group(self, pred_v, faces, foc_crop, pp_crop, rotation, translation, raw=False):
        """mask, depth and canonical-xyz ("hardtex") passes + the projected vertices and their visibility weight:
        (mask_render, depth_render, match_gt, imatch_gt, depth_mask, match_mask, depth_weight).
        raw=True (needs shares_mask_with_depth()): (depth_out [B,4,H,W], match_out [B,4,H,W], imatch_gt, depth_weight) -- the two
        rasteriser outputs undivided, for fused_losses.depth_group_losses."""
        cam = (foc_crop, pp_crop, rotation, translation)
        # The mask pass and the depth pass share sigma, distance function and alpha aggregation, so
        # their alpha planes are bit-identical (checked in tests/test_softras_gpu.py) and the
        # mask pass' gradient reaches the geometry only through alpha: taking the mask from the depth
        # pass' alpha channel is the same forward value and -- by linearity of the backward in the
        # incoming alpha gradient -- the same gradient, with one rasterisation + one backward less.
        fuse_mask = self.shares_mask_with_depth()
        if raw and not fuse_mask:
            raise RuntimeError("render_depth_group(raw=True) needs the mask shared with the depth pass")
        if not fuse_mask:
            mask_render = render(self.renderer_mask, pred_v, faces, None, *cam, render_mask=True)[:, -1]

        # The canonical-xyz ("hardtex") pass rasterises the same projected faces with the same sigma / distance /
        # alpha functions as the depth pass (renderer.py:17-24: only gamma and the rgb aggregation differ), and its
        # inputs are detached values of the same tensors: every coverage decision is shared, so it rides on the depth
        # pass' launch as a second colour output (scp_soft_rasterize_forward_dual; bit-identical images, asserted in
        # tests/test_softras_gpu.py).  It has no backward in the reference either.
        canon = pred_v.detach()
        fuse_hard = pred_v.is_cuda and getattr(self, "share_hardtex_with_depth", True)
        if fuse_hard:
            verts = project_for_render(pred_v, *cam)
            depth_out, match_out = self.renderer_depth.render_mesh_with_hard(
                sr.Mesh(verts, faces, verts, texture_type="vertex"), self.renderer_hardtex, canon)
        else:
            depth_out = render(self.renderer_depth, pred_v, faces, None, *cam, render_depth=True, texture_type="vertex")
            with torch.no_grad():  # zero backward in the reference as well (F8)
                match_out = render(self.renderer_hardtex, canon, faces, canon, *cam, texture_type="vertex")
        if fuse_mask:
            mask_render = depth_out[:, 3]
        if not self.opts.use_depth:
            depth_out = depth_out.detach()
        depth_mask, depth_render = depth_out[:, 3], (depth_out[:, 2] if raw else depth_out[:, 2].clone())
        match_mask, match_gt = match_out[:, -1], match_out[:, :3]

        # projected vertex positions: differentiable w.r.t. rotation / translation (renderer.py:63-67)
        proj = project_vertices(canon, foc_crop, pp_crop, rotation, translation, flip_y=False)          # b,n,3 = (x, y, camera z)
        imatch_gt = proj[:, :, :2].permute(0, 2, 1)   # b,2,n
        with torch.no_grad():  # visibility weight, detached in the reference (:69-71)
            seen = F.grid_sample(depth_render[:, None], proj[:, None, :, :2], align_corners=False)[:, 0, 0]
            depth_weight = (-5 * F.relu(proj[:, :, 2] - seen)).exp()
        if raw:
            return depth_out, match_out, imatch_gt, depth_weight
        return mask_render, depth_render, match_gt, imatch_gt, depth_mask, match_mask, depth_weight

    def render_all(self, pred_v, faces, tex, foc_crop, pp_crop, rotation, translation, scale):
        """the reference's 9-tuple (renderer.py:38-73)"""
        tex_render, tex_mask = self.render_texture(pred_v, faces, tex, foc_crop, pp_crop, rotation, translation)
        (mask_render, depth_render, match_gt, imatch_gt, depth_mask, match_mask,
         depth_weight) = self.render_depth_group(pred_v, faces, foc_crop, pp_crop, rotation, translation)
        return mask_render, tex_render, depth_render, match_gt, imatch_gt, tex_mask, depth_mask, match_mask, depth_weight
