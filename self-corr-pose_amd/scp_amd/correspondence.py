"""scp_amd/correspondence.py -- learned feature<->vertex correspondence.

API and semantics of model/module/correspondence.py: Correspondence.match :36-73 (all-pairs
scores, mask, softmax over pixels and over vertices, soft-argmax both ways, eval-only
forward-backward confidence), compute_rotation_cycle_loss :76-113.  (compute_cycle_loss :116-147 is
never called by the reference and is not provided.)

The all-pairs work goes through scp_amd.ops (fused HIP kernels on the GPU); the [B,HW,V,3]
broadcast temporary of correspondence.py:53 never exists here.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import imgops, ops


def make_meshgrid(hf, wf, device=None):
    """pixel-centre grid in [-1,1]: row 0 = x, row 1 = y, flattened row-major (correspondence.py:31-33)"""
    xs = (torch.arange(wf, dtype=torch.float32, device=device) + 0.5) / (wf / 2) - 1
    ys = (torch.arange(hf, dtype=torch.float32, device=device) + 0.5) / (wf / 2) - 1
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((gx.reshape(-1), gy.reshape(-1)), 0)


class Correspondence:
    def __init__(self, opts, device=None):
        self.opts = opts
        self.tau_img, self.tau_mesh = opts.tau_img, opts.tau_mesh
        self.k_img, self.k_mesh = opts.topk_img, opts.topk_mesh
        self.hf, self.wf = opts.corr_h, opts.corr_w
        self.meshgrid = make_meshgrid(self.hf, self.wf, device)

    def to(self, device):
        self.meshgrid = self.meshgrid.to(device)
        return self

    def match(self, img_feat, mesh_feat, mask, pred_v):
        bsz, h, w = mask.shape
        mask_down = F.interpolate(mask[:, None], (self.hf, self.wf), mode="nearest").reshape(bsz, -1)
        fused = None
        if self.opts.train and img_feat.is_cuda and getattr(self, "fuse_scores", True):
            # training consumes the scores only 2x2-pooled (pretrained_corr.py:120-123): fused kernels, no [B,P,V] tensor
            fused = ops.feature_vertex_match_pooled(img_feat, mesh_feat, mask_down, pred_v.detach(), self.meshgrid,
                                                    self.tau_img, self.tau_mesh, self.hf, self.wf,
                                                    self.half_grid(1).reshape(2, -1) if getattr(self, "fuse_bridge", True) else None)
        if fused is not None:
            pointcorr, match, imatch = fused
        else:
            pointcorr, match, imatch = ops.feature_vertex_match(
                img_feat, mesh_feat, mask_down, pred_v.detach(), self.meshgrid, self.tau_img, self.tau_mesh)

        if self.opts.train:
            match_conf = None
        else:
            with torch.no_grad():
                nearest = ops.nearest_vertex(match, pred_v)                         # b,hw (vertex ids)
                ipred = torch.gather(imatch, 2, nearest[:, None].expand(-1, 2, -1))  # b,2,hw
                fberr = (self.meshgrid[None] - ipred).norm(2, 1).view(bsz, 1, self.hf, self.wf)
                match_conf = F.interpolate((-5 * fberr).exp(), (h, w), mode="bilinear", align_corners=False)
                fg = (mask[:, None] > 0).to(match_conf.dtype)
                conf_mean = torch.clamp((match_conf * fg).sum() / fg.sum(), max=0.5)
                match_conf = torch.where(match_conf < conf_mean, torch.zeros_like(match_conf), match_conf)

        match = F.interpolate(match.reshape(bsz, self.hf, self.wf, 3).permute(0, 3, 1, 2), (h, w), mode="nearest")
        return pointcorr, match, imatch, match_conf

    def half_grid(self, bsz):
        """the constant pixel grid at half resolution (correspondence.py:84-86): interpolated once per device, expanded"""
        key = (self.meshgrid.device, self.meshgrid.data_ptr())
        if getattr(self, "_half_grid_key", None) != key:
            self._half_grid = F.interpolate(self.meshgrid.reshape(2, self.hf, self.wf)[None], (self.hf // 2, self.wf // 2),
                                            mode="bilinear")
            self._half_grid_key = key
            if self._half_grid.is_cuda:
                # cached and read by every stream of the step, but enqueued on whichever stream asked first (SCP_STREAMS=overlap: a side
                # stream): drained once, so that no other stream can read it before it exists (seen as a first-step deviation of the
                # bridge's column soft-argmax, tools/first_step_flake.py)
                torch.cuda.current_stream(self._half_grid.device).synchronize()
        return self._half_grid.expand(bsz, -1, -1, -1)

    def compute_rotation_cycle_loss(self, src_img, src_mask, src_img_feat, encoder, angle=None):
        """`angle` pins the random rotation (tests); None draws U(0,360) from the host RNG"""
        bsz, c = src_img.shape[0], self.opts.n_corr_feat
        if angle is None:
            angle = torch.empty(1).uniform_(0., 360.).item()
        hh, wh = self.hf // 2, self.wf // 2
        grid = self.half_grid(bsz)
        src_mask = src_mask[:, None]
        tgt_img = imgops.rotate(src_img, angle, "bilinear")
        tgt_mask = imgops.rotate(src_mask, angle, "nearest")
        # every image gets the same rotation of the same grid: rotate one copy
        cycle_match_gt = imgops.rotate(grid[:1], angle, "nearest").expand(bsz, -1, -1, -1).reshape(bsz, 2, -1)

        src_mask_down = F.interpolate(src_mask, (hh, wh), mode="nearest").reshape(bsz, -1)
        tgt_mask_down = F.interpolate(tgt_mask, (hh, wh), mode="nearest").reshape(bsz, -1)
        if getattr(encoder, "supports_half_res", False) and self.hf % 2 == 0 and self.wf % 2 == 0:
            # nearest down-sampling by exactly 2 keeps the even pixels: ask the encoder for those only (the last decoder
            # unit and the projection then run on a quarter of the pixels; same values)
            _, tgt_half = encoder.encode_img(tgt_img, half_res=True)                # b,c,hh*wh (unit norm)
        else:
            _, tgt_feat = encoder.encode_img(tgt_img)                               # b,c,hf*wf (unit norm)
            tgt_half = F.interpolate(tgt_feat.reshape(bsz, c, self.hf, self.wf), (hh, wh), mode="nearest").reshape(bsz, c, -1)
        src_half = F.interpolate(src_img_feat.reshape(bsz, c, self.hf, self.wf), (hh, wh), mode="nearest").reshape(bsz, c, -1)

        cycle_match = ops.pixel_pixel_softargmax(src_half, tgt_half, src_mask_down, tgt_mask_down,
                                                 grid.reshape(bsz, 2, -1), self.tau_mesh)
        cycle_loss = ((cycle_match - cycle_match_gt).norm(2, 1) * tgt_mask_down).mean()
        return cycle_loss, cycle_match, cycle_match_gt, tgt_mask_down
