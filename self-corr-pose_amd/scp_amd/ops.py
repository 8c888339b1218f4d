"""scp_amd/ops.py -- the dense correspondence operators of the step, behind one small interface.

Each function states its tensor contract once; Correspondence / PretrainedCorrespondence call these
and nothing else for all-pairs work.  The reductions that have hand-written HIP kernels
(cols_softargmax, feature_vertex_match, mutual_nn; csrc/corr.hip, csrc/mutual_nn.hip via scp_amd.corr_ops) accept GPU tensors ONLY and
raise otherwise -- there is no CPU fallback in the product; CPU tests substitute the oracle
(oracle/corr.py) for them with monkeypatch (tests/oracle_backend.py).  The remaining functions are
compositions of stock PyTorch ops (library GEMMs, argmax, gather) and run on any device.

Reference call sites: model/module/correspondence.py:42-53 (feature_vertex_match), :58-60
(nearest_vertex), :105-110 (pixel_pixel_softargmax); model/module/pretrained_corr.py:85-102
(mutual_nn, mutual_nn_pairs), :120-137 (pool2x2_scores, vertex_bridge_match).
"""
import torch

from . import corr_ops


def _require_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError("scp_amd.ops.%s runs on the HIP kernels only (got a %s tensor; no CPU fallback)" % (name, t.device))


def _masked(pc, keep):
    """pc * [m>0] - 1e5 * [m==0]  (correspondence.py:44, pretrained_corr.py:86)"""
    return torch.where(keep, pc, torch.full_like(pc, -1e5))


def cols_softargmax(scores, rowmask, colmask, grid, tau):
    """scores [N,P,Q]; rowmask [N,P] / colmask [N,Q] or None (entry masked to -1e5 where a mask is 0);
    grid [2,P] or [N,2,P]  ->  [N,2,Q] = grid @ softmax_P(tau * masked(scores))"""
    _require_gpu(scores, "cols_softargmax")
    return corr_ops.ColsSoftArgmax.apply(scores, rowmask, colmask, grid, tau)


def feature_vertex_match(img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh):
    """img_feat [B,C,P], mesh_feat [B,V,C], mask_down [B,P], verts [B,V,3] (no grad), grid [2,P]
    -> pointcorr [B,P,V] (masked scores), match [B,P,3] = softmax_V(tau_img*pc) @ verts,
       imatch [B,2,V] = grid @ softmax_P(tau_mesh*pc)"""
    _require_gpu(img_feat, "feature_vertex_match")
    return corr_ops.FeatureVertexMatch.apply(img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh)


def feature_vertex_match_pooled(img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh, hf, wf, grid_half=None):
    """training form of feature_vertex_match: the scores are consumed only 2x2-pooled (pretrained_corr.py:120-123), so they
    are produced that way and never stored at full resolution.  -> (PooledScores, match [B,P,3], imatch [B,2,V]);
    None when the shape is outside what the fused kernels cover (caller falls back to feature_vertex_match).
    grid_half [2,P/4] (the pixel grid at the pooled resolution): the same launch also leaves, in PooledScores.bridge, the column
    soft-argmax of the pooled scores that the vertex bridge needs (pretrained_corr.py:123-126; vertex_bridge_match(precomputed=...))."""
    _require_gpu(img_feat, "feature_vertex_match_pooled")
    if not corr_ops.fvm_fusable(img_feat, mesh_feat, hf, wf):
        return None
    if grid_half is None:
        pooled, match, imatch = corr_ops.FeatureVertexMatchFused.apply(img_feat, mesh_feat, mask_down, verts, grid, tau_img,
                                                                       tau_mesh, hf, wf)
        return corr_ops.PooledScores(pooled, hf, wf), match, imatch
    pooled, match, imatch, xy, stat = corr_ops.FeatureVertexMatchFused.apply(img_feat, mesh_feat, mask_down, verts, grid, tau_img,
                                                                             tau_mesh, hf, wf, grid_half)
    return corr_ops.PooledScores(pooled, hf, wf, (grid_half, float(tau_mesh), xy, stat)), match, imatch


def nearest_vertex(points, verts):
    """points [B,P,3], verts [B,V,3] -> index [B,P] of the nearest vertex (L2); eval only"""
    d = points.pow(2).sum(-1)[:, :, None] - 2 * points.bmm(verts.transpose(1, 2)) + verts.pow(2).sum(-1)[:, None, :]
    return d.argmin(2)


def pixel_pixel_softargmax(src_feat, tgt_feat, src_mask, tgt_mask, grid, tau):
    """src/tgt_feat [B,C,P], masks [B,P], grid [B,2,P] -> [B,2,P_tgt] = grid @ softmax_src(tau*pc)
    with pc = src^T tgt masked by src_mask x tgt_mask"""
    if corr_ops.pp_fusable(src_feat, tgt_feat):           # the training shapes: scores stay in registers (csrc/corr_pp.hip)
        return corr_ops.PixelPixelSoftArgmax.apply(src_feat, tgt_feat, src_mask, tgt_mask, grid, tau)
    pc = src_feat.transpose(1, 2).bmm(tgt_feat)
    return cols_softargmax(pc, src_mask, tgt_mask, grid, tau)


def mutual_nn(src_feat, tgt_feat, src_mask, tgt_mask):
    """src/tgt_feat [N,C,P] -> (bw [N,P_tgt] = argmax over src, fw [N,P_src] = argmax over tgt) of the
    masked score matrix; ties resolve to the lowest index.  One pass over the score tensor (csrc/mutual_nn.hip)."""
    _require_gpu(src_feat, "mutual_nn")
    pc = src_feat.transpose(1, 2).bmm(tgt_feat)
    return corr_ops.mutual_argmax(pc, src_mask, tgt_mask)


def mutual_nn_pairs(keys, src_img, tgt_img, mask_down, tok0=1):
    """The training path's form of mutual_nn (pretrained_corr.py:59-89 in one kernel, csrc/mutual_nn.hip scp_mutual_nn_fused): keys
    [n_images, n_tok, C] token-major DINO keys (class token at row 0 .. tok0 - 1), src_img / tgt_img [N] the images of each pair,
    mask_down [n_images, P] per image  ->  (bw [N,P], fw [N,P]) as mutual_nn; neither the gathered per-pair feature maps nor the
    [N,P,P] score tensor exist.  The products run on the matrix cores the ViT's linear layers use (scp_amd.dino.GEMM_MODE)."""
    _require_gpu(keys, "mutual_nn_pairs")
    from . import dino
    mode = dino.gemm_mode()
    planes = getattr(keys, "_scp_planes", None)            # left by the K projection's epilogue (scp_amd/dino.py) when it pre-splits
    if mode == "fp32":
        planes = "fp32"
    elif planes is None or (planes.rows, planes.cols) != (keys.shape[0] * keys.shape[1], keys.shape[2]):
        planes = "split"
    return corr_ops.mutual_nn_fused(keys, src_img, tgt_img, mask_down, tok0, planes)


def pool2x2_scores(pc, hf, wf):
    """pc [B, hf*wf, V] -> [B, (hf/2)*(wf/2), V]: 2x2 spatial mean of every vertex' score map.
    The reference reaches it through a bilinear F.interpolate to half resolution of the permuted
    [B,V,hf,wf] view (pretrained_corr.py:120-123), which for an exact factor 2 with
    align_corners=False is this mean."""
    if isinstance(pc, corr_ops.PooledScores):          # produced pooled by the fused kernels (feature_vertex_match_pooled)
        assert (pc.hf, pc.wf) == (hf, wf)
        return pc.pooled
    b, _, v = pc.shape
    # the association of ATen's bilinear kernel for an exact factor 2 -- 0.5 * (0.5 a + 0.5 b) + 0.5 * (0.5 c + 0.5 d) with (a, b) the
    # horizontal neighbours: scalings by powers of two are exact, so this is 0.25 * ((a + b) + (c + d)) bit for bit.  The order matters
    # at the mask border, where a cell mixes -1e5 with real scores and the sum is rounded at 2^-7 (the G4 fixture sees it per entry).
    x = pc.reshape(b, hf // 2, 2, wf // 2, 2, v)
    return (0.25 * ((x[:, :, 0, :, 0] + x[:, :, 0, :, 1]) + (x[:, :, 1, :, 0] + x[:, :, 1, :, 1]))).reshape(b, -1, v)


def vertex_bridge_match(pooled, src_idx, tgt_idx, tgt_pixels, keep, grid_half, tau_img, tau_mesh, precomputed=None):
    """Soft pixel->pixel map through the vertices, evaluated at the selected target pixels.
    pooled [B,P,V] per-image pooled scores; src_idx/tgt_idx [N] image pairs; tgt_pixels [N,K] selected
    target pixels; keep [B,V] bool (vertex visible); grid_half [2,P]  ->  [N,2,K].

    Reference (pretrained_corr.py:123-137):  corr = (softmax_P(tau_mesh*pc_src)*keep_src) @
    (softmax_V(tau_img*pc_tgt)*keep_tgt)^T, columns normalised by (sum_P + 1e-5), match = grid @ corr,
    then gathered at tgt_pixels.  Re-associated, exactly the same sums:
        grid @ corr[:, :, k]  =  sum_v keep_src[v] keep_tgt[v] P_img[k,v] * (grid @ P_mesh)[:, v]
        sum_P corr[:, :, k]   =  sum_v keep_src[v] keep_tgt[v] P_img[k,v] * (sum_p P_mesh[p,v] = 1)
    so the [N,P,P] matrix and its 86-134 GFLOP GEMM are never formed; (grid @ P_mesh) is one column
    soft-argmax per unique image."""
    num_verts = pooled.shape[-1]
    # the K selected rows of each pair's target image, taken straight from pooled[B,P/4,V] (one row gather forward, one row
    # scatter-add backward): no per-pair [N,P/4,V] copy of the pooled scores (168 MB at N = 64) and none of its gradient
    n_pool = pooled.shape[1]
    rows = (tgt_idx[:, None] * n_pool + tgt_pixels).reshape(-1)
    if precomputed is not None and precomputed[1] == float(tau_mesh) and precomputed[0].shape == grid_half.shape:
        # PooledScores.bridge: the kernel that produced `pooled` already took the column soft-argmax (same grid values by
        # construction: both sides interpolate make_meshgrid the same way); one autograd node for both reads of `pooled`
        sel, mxy = corr_ops.BridgeInputs.apply(pooled, rows, precomputed[0], tau_mesh, precomputed[2], precomputed[3])
    else:
        mxy = cols_softargmax(pooled, None, None, grid_half, tau_mesh)                   # [B,2,V]
        sel = pooled.reshape(-1, num_verts).index_select(0, rows)
    pc_tgt_sel = sel.reshape(tgt_pixels.shape[0], tgt_pixels.shape[1], num_verts)
    p_img = torch.softmax(tau_img * pc_tgt_sel, dim=2)                                   # [N,K,V]
    both = (keep[src_idx] & keep[tgt_idx]).to(pooled.dtype)                              # [N,V]
    p_img = p_img * both[:, None, :]
    num = mxy[src_idx].bmm(p_img.transpose(1, 2))                                        # [N,2,K]
    den = p_img.sum(2)[:, None, :] + 1e-5
    return num / den
