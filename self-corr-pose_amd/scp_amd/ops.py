"""scp_amd/ops.py -- the dense correspondence operators of the step, behind one small interface.

Each function states the tensor contract once; Correspondence / PretrainedCorrespondence call these
and nothing else for all-pairs work.  GPU tensors run the hand-written HIP kernels of
csrc/ (through scp_amd.capi); the formulations below in plain torch ops are the exact definition of
every operator and run for non-GPU tensors (host-logic tests).

Reference call sites: model/module/correspondence.py:42-53 (feature_vertex_match), :58-60
(nearest_vertex), :105-110 (pixel_pixel_softargmax); model/module/pretrained_corr.py:85-102
(mutual_nn_topk), :120-137 (vertex_bridge_match).
"""
import torch


def _masked(pc, keep):
    """pc * [m>0] - 1e5 * [m==0]  (correspondence.py:44, pretrained_corr.py:86)"""
    return torch.where(keep, pc, torch.full_like(pc, -1e5))


def feature_vertex_match(img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh):
    """img_feat [B,C,P], mesh_feat [B,V,C], mask_down [B,P], verts [B,V,3] (no grad), grid [2,P]
    -> pointcorr [B,P,V] (masked scores), match [B,P,3] = softmax_V(tau_img*pc) @ verts,
       imatch [B,2,V] = grid @ softmax_P(tau_mesh*pc)"""
    pc = mesh_feat.bmm(img_feat).permute(0, 2, 1)
    pc = _masked(pc, (mask_down > 0)[:, :, None])
    p_mesh = torch.softmax(tau_mesh * pc, dim=1)
    p_img = torch.softmax(tau_img * pc, dim=2)
    imatch = grid[None].expand(pc.shape[0], -1, -1).bmm(p_mesh)
    match = p_img.bmm(verts)
    return pc, match, imatch


def nearest_vertex(points, verts):
    """points [B,P,3], verts [B,V,3] -> index [B,P] of the nearest vertex (L2)"""
    d = points.pow(2).sum(-1)[:, :, None] - 2 * points.bmm(verts.transpose(1, 2)) + verts.pow(2).sum(-1)[:, None, :]
    return d.argmin(2)


def pixel_pixel_softargmax(src_feat, tgt_feat, src_mask, tgt_mask, grid, tau):
    """src/tgt_feat [B,C,P], masks [B,P], grid [B,2,P] -> [B,2,P_tgt] = grid @ softmax_src(tau*pc)
    with pc = src^T tgt masked by src_mask x tgt_mask"""
    pc = src_feat.transpose(1, 2).bmm(tgt_feat)
    keep = (src_mask[:, :, None] * tgt_mask[:, None, :]) > 0
    return grid.bmm(torch.softmax(tau * _masked(pc, keep), dim=1))


def mutual_nn(src_feat, tgt_feat, src_mask, tgt_mask):
    """src/tgt_feat [N,C,P] -> (bw [N,P_tgt] = argmax over src, fw [N,P_src] = argmax over tgt) of the
    masked score matrix; ties resolve to the lowest index"""
    pc = src_feat.transpose(1, 2).bmm(tgt_feat)
    pc = _masked(pc, (src_mask[:, :, None] * tgt_mask[:, None, :]) > 0)
    return pc.max(1).indices, pc.max(2).indices


def vertex_bridge_match(pc_src, pc_tgt_sel, keep_src, keep_tgt, grid, tau_img, tau_mesh):
    """soft pixel->pixel map through the vertices, evaluated only at the selected target pixels.
    pc_src [N,P,V] (all source pixels), pc_tgt_sel [N,K,V] (selected target pixels),
    keep_src/keep_tgt [N,V] bool (vertex visible), grid [N,2,P]
    -> [N,2,K]: grid @ normalise_columns( softmax_P(tau_mesh*pc_src)*keep_src @ (softmax_V(tau_img*pc_tgt_sel)*keep_tgt)^T )
    (pretrained_corr.py:123-137; a column of `corr` depends only on its own target pixel, so
    restricting to the K gathered columns is exact)"""
    p_mesh = torch.softmax(tau_mesh * pc_src, dim=1) * keep_src[:, None, :]
    p_img = torch.softmax(tau_img * pc_tgt_sel, dim=2) * keep_tgt[:, None, :]
    corr = p_mesh.bmm(p_img.transpose(1, 2))                       # N,P,K
    corr = corr / (corr.sum(1, keepdim=True) + 1e-5)
    return grid.bmm(corr)


def pool2x2_scores(pc, hf, wf):
    """pc [B, hf*wf, V] -> [B, (hf/2)*(wf/2), V]: 2x2 spatial mean of every vertex' score map.
    The reference reaches it through a bilinear F.interpolate to half resolution of the permuted
    [B,V,hf,wf] view (pretrained_corr.py:120-123), which for an exact factor 2 with
    align_corners=False is this mean (weights 1/4 each; last-ulp rounding order differs)."""
    b, _, v = pc.shape
    x = pc.reshape(b, hf // 2, 2, wf // 2, 2, v)
    return ((x[:, :, 0, :, 0] + x[:, :, 0, :, 1]) * 0.5 * 0.5 + (x[:, :, 1, :, 0] + x[:, :, 1, :, 1]) * 0.5 * 0.5).reshape(b, -1, v)
