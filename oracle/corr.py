"""oracle/corr.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement (plain torch-CPU ops, written the way the reference writes them: every softmax and
all-pairs tensor materialised) of the dense correspondence operators.  Only tests/, smoke() and the
cpu_baseline leg of bench.py may import it.  Reference lines (under /root/reference/model/module/):
  match_oracle               correspondence.py:36-53
  rotation_cycle_oracle      correspondence.py:100-112 (the part after the encoder)
  mutual_nn_topk_oracle      pretrained_corr.py:80-104
  bridge_cycle_oracle        pretrained_corr.py:118-139
Pinned by tests/golden/corr_*.npz (recorded from the reference's own classes, make_golden.py gen_corr)
and end-to-end by the full-step fixture.
"""
import torch
import torch.nn.functional as F


def match_oracle(img_feat, mesh_feat, mask_down, pred_v, meshgrid, tau_img, tau_mesh):
    bsz = img_feat.shape[0]
    pointcorr = mesh_feat.bmm(img_feat)                                   # b,n,h*w
    pointcorr = pointcorr.permute(0, 2, 1)                                # b,h*w,n
    pointcorr = pointcorr * (mask_down[:, :, None] > 0) - 1e5 * (mask_down[:, :, None] == 0)
    pointcorr_mesh = torch.softmax(tau_mesh * pointcorr, dim=1)
    pointcorr_img = torch.softmax(tau_img * pointcorr, dim=2)
    imatch = meshgrid[None].repeat(bsz, 1, 1).bmm(pointcorr_mesh)         # b,2,n
    match = (pointcorr_img[:, :, :, None] * pred_v.detach()[:, None, :, :]).sum(2)   # b,h*w,3
    return pointcorr, match, imatch


def rotation_cycle_oracle(src_feat_half, tgt_feat_half, src_mask_down, tgt_mask_down, grid, cycle_match_gt, tau):
    mask_down = src_mask_down[:, :, None] * tgt_mask_down[:, None, :]
    pointcorr = src_feat_half.permute(0, 2, 1).bmm(tgt_feat_half)
    pointcorr = pointcorr * (mask_down > 0) - 1e5 * (mask_down == 0)
    pointcorr_tgt = torch.softmax(tau * pointcorr, dim=1)
    cycle_match = grid.bmm(pointcorr_tgt)
    cycle_loss = ((cycle_match - cycle_match_gt).norm(2, 1) * tgt_mask_down).mean()
    return cycle_loss, cycle_match


def mutual_nn_oracle(src_feat, tgt_feat, src_mask_down, tgt_mask_down):
    mask_down = src_mask_down[:, :, None] * tgt_mask_down[:, None, :]
    pointcorr = src_feat.permute(0, 2, 1).bmm(tgt_feat)
    pointcorr = pointcorr * (mask_down > 0) - 1e5 * (mask_down == 0)
    return pointcorr.max(1).indices, pointcorr.max(2).indices, pointcorr


def bridge_cycle_oracle(pointcorr_src, pointcorr_tgt, dw_src, dw_tgt, grid, indices_tgt, pts_src, mask, hf, wf,
                        tau_img, tau_mesh):
    """pointcorr_* [n, hf*wf, V] full-resolution masked scores of the paired images"""
    bsz, _, num_verts = pointcorr_src.shape
    hh, wh = hf // 2, wf // 2

    def half(pc):
        return F.interpolate(pc.permute(0, 2, 1).reshape(bsz, num_verts, hf, wf), (hh, wh),
                             mode="bilinear").reshape(bsz, num_verts, hh * wh).permute(0, 2, 1)

    pointcorr_src, pointcorr_tgt = half(pointcorr_src), half(pointcorr_tgt)
    pointcorr_img = torch.softmax(tau_img * pointcorr_tgt, dim=2)
    pointcorr_mesh = torch.softmax(tau_mesh * pointcorr_src, dim=1)
    pointcorr_img = pointcorr_img * (dw_tgt[:, None] >= 0.5)
    pointcorr_mesh = pointcorr_mesh * (dw_src[:, None] >= 0.5)
    corr = pointcorr_mesh.bmm(pointcorr_img.permute(0, 2, 1))
    corr = corr / (corr.sum(1, keepdims=True) + 1e-5)
    match = grid.bmm(corr)
    match = torch.gather(match, -1, indices_tgt[:, None].repeat(1, 2, 1))
    cycle_loss = ((match - pts_src).norm(2, 1) * mask).mean()
    return cycle_loss, match


def cols_softargmax_oracle(scores, rowmask, colmask, grid, tau):
    """grid @ softmax_P(tau * masked(scores)) with the reference's masking expression
    (correspondence.py:44 / :106 / pretrained_corr.py:86): scores [N,P,Q], masks [N,P] / [N,Q] or None"""
    keep = torch.ones_like(scores)
    if rowmask is not None:
        keep = keep * rowmask[:, :, None]
    if colmask is not None:
        keep = keep * colmask[:, None, :]
    pc = scores * (keep > 0) - 1e5 * (keep == 0)
    g = grid if grid.dim() == 3 else grid[None].repeat(scores.shape[0], 1, 1)
    return g.bmm(torch.softmax(tau * pc, dim=1))


def nearest_index_oracle(x, y):
    """pytorch3d.ops.knn_points(x, y, K=1).idx[..., 0] (model/util/chamfer.py:135): exact squared
    distances, lowest index on ties"""
    return (x[:, :, None] - y[:, None]).pow(2).sum(-1).argmin(-1)
