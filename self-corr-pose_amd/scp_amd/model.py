"""scp_amd/model.py -- MeshNet: one self-supervised training forward.

API of model/model.py (MeshNet.__init__ :44-58, forward :61-152,309, load_network :313-328): the
same 12-tuple in, `(total_loss, aux_output)` out in training and the same 10-tuple in eval; the same
sub-module names (mesh, encoder, pretrain_corr_net, triangle_loss_fn) so checkpoints and the
optimiser's name-based parameter groups carry over.  The per-`vis_freq` visualisation block
(model.py:155-307: host-side drawing with cv2/trimesh) is not part of the hot path and is omitted.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dino, fused_losses, losses, streams
from .correspondence import Correspondence
from .encoder import Encoder
from .mesh import CanonicalMesh
from .pretrained_corr import PretrainedCorrespondence
from .renderer import Renderer
from .weights import Weights


class MeshNet(nn.Module):
    def __init__(self, opts, prior=None):
        super().__init__()
        self.opts = opts
        self.mesh = CanonicalMesh(opts, prior)
        self.weights = Weights(opts)
        self.encoder = Encoder(opts)
        self.corr_net = Correspondence(opts)
        self.pretrain_corr_net = PretrainedCorrespondence(opts, self.mesh, pretrained=True)
        self.renderer = Renderer(opts, self.mesh)
        self.iters = 0
        self.triangle_loss_fn = losses.LaplacianLoss(self.mesh.mean_v, self.mesh.faces, average=True)
        if opts.flatten_loss:
            self.flatten_loss_fn = losses.FlattenLoss(self.mesh.faces, average=True)
        self.rotation_angle = None   # test hook: pins the rotation-cycle angle

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.corr_net.to(self.mesh.mean_v.device)   # Correspondence is a plain object holding a grid
        return out

    def forward(self, data):
        opts, wts = self.opts, self.weights
        dino.MIXED_BF16 = bool(getattr(opts, "mixed_bf16", False))
        wts.schedule(self.iters)
        img, mask, depth, occ, center, length, foc, foc_crop, pp, pp_crop, indices, gt = data
        bsz = img.shape[0]
        mean_v = self.mesh.mean_v[None].expand(bsz, -1, -1)
        faces = self.mesh.faces[None].expand(bsz, -1, -1)

        if opts.train and getattr(self, "overlap_dino", streams.overlap()):
            self.pretrain_corr_net.prefetch_features(img, mask)
        img_feat, mesh_feat, pred_v, rotation, translation, scale = self.encoder(img, mean_v, pp_crop, foc_crop)
        streams.crumb("forward: encoder done")
        # The rotation-cycle branch (a second, independent encoder pass over the rotated images) only
        # needs img / mask / img_feat: it runs on a side HIP stream next to the correspondence + render +
        # loss work of the main stream (many small, latency-bound kernels); autograd replays each
        # backward node on its forward stream, so the two backward halves overlap as well.
        cycle_side = None
        if opts.train and img.is_cuda and getattr(self, "overlap_rotation_cycle", streams.overlap()):
            if getattr(self, "_cycle_stream", None) is None:
                self._cycle_stream = streams.side_stream(img.device)
            main = torch.cuda.current_stream(img.device)
            self._cycle_stream.wait_stream(main)
            with torch.cuda.stream(self._cycle_stream):
                streams.crumb("rotation cycle: start")
                cycle_side = self.corr_net.compute_rotation_cycle_loss(img, mask, img_feat, self.encoder,
                                                                       angle=self.rotation_angle)[0]
                streams.crumb("rotation cycle: done")
            for t in (img, mask, img_feat):
                t.record_stream(self._cycle_stream)

        pointcorr, match, imatch, match_conf = self.corr_net.match(img_feat, mesh_feat, mask, pred_v)
        tex = self.mesh.get_texture(pred_v, faces, imatch, img)
        if not opts.train:
            return pred_v, faces, tex, imatch, match, match_conf, rotation, translation, scale, pointcorr

        occ_arg = occ if opts.use_occ else None
        cam = (foc_crop, pp_crop, rotation, translation)
        # The soft-texture pass and its loss share nothing with the mask / depth / canonical-xyz group but their inputs: with
        # SCP_TEXTURE_STREAM=1 they run on a side stream of their own (off by default since round 6, streams.overlap_texture()).  Both chains are rasteriser launches (VALU-bound, far from filling the device)
        # strung together by small latency-bound kernels, forward and -- autograd replays nodes on their forward stream --
        # backward; side by side they shorten the step's serial middle part.
        tex_side = opts.train and img.is_cuda and tex is not None and getattr(self, "overlap_texture_pass", streams.overlap_texture())
        if tex_side:
            if getattr(self, "_tex_stream", None) is None:
                self._tex_stream = streams.side_stream(img.device)
            main = torch.cuda.current_stream(img.device)
            self._tex_stream.wait_stream(main)
            with torch.cuda.stream(self._tex_stream):
                streams.crumb("texture pass: start")
                texture_loss = wts.tex_wt * self._texture_loss(pred_v, faces, tex, cam, img, mask, occ_arg).mean(0)
                streams.crumb("texture pass: done")
            for t in (pred_v, faces, tex, foc_crop, pp_crop, rotation, translation, img, mask) + ((occ_arg,) if occ_arg is not None else ()):
                t.record_stream(self._tex_stream)
        else:
            texture_loss = wts.tex_wt * self._texture_loss(pred_v, faces, tex, cam, img, mask, occ_arg).mean(0)
        # fused per-pixel losses (csrc/losses.hip) when they cover the configuration, the torch compositions otherwise
        fused = (getattr(self, "fuse_image_losses", True) and occ_arg is None and opts.use_depth and not opts.depth_loss_chamfer
                 and self.renderer.shares_mask_with_depth() and fused_losses.covers(img, mask, depth, match))
        if fused:
            depth_out, match_out, imatch_gt, depth_weight = self.renderer.render_depth_group(pred_v, faces, *cam, raw=True)
            mask_sub, depth_sub, match_sub = fused_losses.depth_group_losses(depth_out, match_out, match, depth, mask)
            mask_loss, depth_loss = wts.mask_wt * mask_sub.mean(0), wts.depth_wt * depth_sub.mean(0)
            match_loss = wts.match_wt * match_sub.mean(0)
        else:
            (mask_render, depth_render, match_gt, imatch_gt, depth_mask, match_mask,
             depth_weight) = self.renderer.render_depth_group(pred_v, faces, *cam)
            mask_loss = wts.mask_wt * losses.compute_mask_loss(img, mask, mask_render, occ_arg).mean(0)
            if opts.use_depth:
                if opts.depth_loss_chamfer:
                    raise NotImplementedError("depth_loss_chamfer is off in every shipped config")
                depth_loss_sub, _ = losses.compute_depth_loss(depth, depth_render, depth_mask, mask)
                depth_loss = wts.depth_wt * depth_loss_sub.mean(0)
            match_loss = wts.match_wt * losses.compute_match_loss(match, match_gt, match_mask, mask).mean(0)
        imatch_loss = wts.imatch_wt * losses.compute_imatch_loss(imatch, imatch_gt, depth_weight).mean(0)

        symmetry_loss = wts.symmetry_wt * self.mesh.compute_symmetry_loss(pred_v, faces)
        triangle_loss = wts.triangle_wt * self.triangle_loss_fn(pred_v) * pred_v.shape[1] / 64.
        if opts.flatten_loss:
            triangle_loss = triangle_loss + wts.triangle_wt * self.flatten_loss_fn(pred_v) * 0.1 * np.sqrt(pred_v.shape[1] / 64.)
        pullfar_loss = wts.pullfar_wt * F.relu(1 - translation[:, :, -1]).mean()
        deform_loss = wts.deform_wt * F.smooth_l1_loss(pred_v, mean_v, reduction="mean")

        streams.crumb("forward: render + image losses + regularisers done")
        cycle_loss_pt = self.pretrain_corr_net.compute_cycle_loss(img, mask, depth_weight, pointcorr)[0] * wts.cycle_loss_pt_wt
        streams.crumb("forward: ViT features joined, bridge loss done")
        if cycle_side is not None:
            torch.cuda.current_stream(img.device).wait_stream(self._cycle_stream)
            cycle_loss = cycle_side * wts.cycle_loss_wt
        else:
            cycle_loss = self.corr_net.compute_rotation_cycle_loss(img, mask, img_feat, self.encoder,
                                                                   angle=self.rotation_angle)[0] * wts.cycle_loss_wt

        if tex_side:
            torch.cuda.current_stream(img.device).wait_stream(self._tex_stream)
            texture_loss.record_stream(torch.cuda.current_stream(img.device))
        total_loss = (mask_loss + symmetry_loss + triangle_loss + deform_loss + pullfar_loss + texture_loss +
                      match_loss + imatch_loss + cycle_loss_pt + cycle_loss)
        if opts.use_depth:
            total_loss = total_loss + depth_loss
        if opts.camera_loss:
            rot2 = rotation.detach().reshape(-1, opts.repeat, 3, 3).roll(-1, 1).reshape(bsz, 3, 3)
            cam_loss = wts.camera_wt * losses.compute_camera_loss(rotation, rot2).mean()
            total_loss = total_loss + cam_loss

        aux_output = {
            "total_loss": total_loss, "mask_loss": mask_loss, "triangle_loss": triangle_loss,
            "deform_loss": deform_loss, "pullfar_loss": pullfar_loss, "symmetry_loss": symmetry_loss,
            "match_loss": match_loss, "texture_loss": texture_loss, "imatch_loss": imatch_loss,
            "cycle_loss_pretrain": cycle_loss_pt, "cycle_loss": cycle_loss,
        }
        if opts.use_depth:
            aux_output["depth_loss"] = depth_loss
        if opts.camera_loss:
            aux_output["cam_loss"] = cam_loss
        self.last_pose = (rotation.detach(), translation.detach())
        # the encoder's outputs of this step (parity runs pin them on the other side; references, no copies)
        self.last_geometry = (pred_v.detach(), rotation.detach(), translation.detach())
        self.last_features = (img_feat.detach(), mesh_feat.detach())
        return total_loss, aux_output

    def _texture_loss(self, pred_v, faces, tex, cam, img, mask, occ):
        """soft-texture pass + texture loss [B]"""
        if tex is not None and occ is None and getattr(self, "fuse_image_losses", True) and fused_losses.covers(img, mask):
            return fused_losses.texture_loss(self.renderer.render_texture_raw(pred_v, faces, tex, *cam), img, mask)
        tex_render, tex_mask = self.renderer.render_texture(pred_v, faces, tex, *cam)
        return losses.compute_texture_loss(img, mask, tex_render, tex_mask, occ)

    def load_network(self, model_path, iter=0):
        states = torch.load(model_path, map_location="cpu")
        for name in list(states.keys()):
            if "symm_rots" in name or "triangle_loss_fn" in name or "flatten_loss_fn" in name:
                states.pop(name)
        self.load_state_dict(states, strict=False)
        from . import fused_conv
        fused_conv.WEIGHT_EPOCH[0] += 1          # cached operand planes of the convolution weights are stale now
