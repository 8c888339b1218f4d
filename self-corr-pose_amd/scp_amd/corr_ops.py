"""scp_amd/corr_ops.py -- autograd nodes over the HIP correspondence kernels (csrc/corr.hip, C ABI in
include/scp_hip.h).  GPU tensors only; scp_amd.ops routes here."""
import ctypes

import torch
from torch.autograd import Function

from . import capi


def _ws(n, p, q, device):
    nbytes = capi.lib().scp_softargmax_cols_workspace(n, p, q)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def cols_forward(scores, rowmask, colmask, grid, tau, masked_out=None):
    n, p, q = scores.shape
    out = torch.empty(n, 2, q, dtype=torch.float32, device=scores.device)
    stats = torch.empty(n, 2, q, dtype=torch.float32, device=scores.device)
    ws, nbytes = _ws(n, p, q, scores.device)
    code = capi.lib().scp_softargmax_cols_forward(
        capi.dev_ptr(scores, "scores"), capi.opt_ptr(masked_out, "scores_masked_out"),
        capi.opt_ptr(rowmask, "rowmask"), capi.opt_ptr(colmask, "colmask"), capi.dev_ptr(grid, "grid"),
        int(grid.dim() == 3), float(tau), n, p, q, capi.dev_ptr(out, "out"), capi.dev_ptr(stats, "colstats"),
        capi.dev_ptr(ws, "workspace"), ctypes.c_size_t(nbytes), capi.current_stream())
    capi.check(code, "scp_softargmax_cols_forward")
    return out, stats


def mutual_argmax(scores, rowmask, colmask):
    """scores [N,P,Q] -> (col_index [N,Q] = argmax over rows, row_index [N,P] = argmax over columns) of the masked
    matrix (entries with either mask <= 0 count as -1e5), lowest index on ties; no gradient"""
    L = capi.lib()
    n, p, q = scores.shape
    scores = scores.detach().contiguous().float()
    rm = None if rowmask is None else rowmask.contiguous().float()
    cm = None if colmask is None else colmask.contiguous().float()
    col = torch.empty(n, q, dtype=torch.int64, device=scores.device)
    row = torch.empty(n, p, dtype=torch.int64, device=scores.device)
    nbytes = L.scp_mutual_argmax_workspace(n, q)
    ws = torch.empty(nbytes // 8, dtype=torch.int64, device=scores.device)
    capi.check(L.scp_mutual_argmax(capi.dev_ptr(scores, "scores"), capi.opt_ptr(rm, "rowmask"), capi.opt_ptr(cm, "colmask"),
                                   n, p, q, ctypes.c_void_p(col.data_ptr()), ctypes.c_void_p(row.data_ptr()),
                                   ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nbytes), capi.current_stream()),
               "scp_mutual_argmax")
    return col, row


def mutual_nn_fused(keys, src_img, tgt_img, mask_down, tok0=1, planes="split"):
    """scp_mutual_nn_fused: keys [n_images, n_tok, C] fp32 token-major (DINO.key_tokens), src_img / tgt_img [N] image indices of
    the pairs, mask_down [n_images, P] (P = n_tok - tok0) or None  ->  (bw [N,P] = best source token per target token, fw [N,P] =
    best target token per source token) of the masked score matrix S = K_src K_tgt^T, which is never stored.  `planes`: "split"
    (bf16 matrix cores on exactly split operands; the planes are made here) | "fp32" (fp32 matrix cores) | the dino.TiledPlanes
    [n_images * n_tok, C] of the keys the caller already holds (the K projection's epilogue writes them)."""
    L = capi.lib()
    if not (keys.is_cuda and keys.dtype == torch.float32 and keys.dim() == 3):
        raise RuntimeError("mutual_nn_fused: keys must be a CUDA float32 [n_images, n_tok, C] tensor (no CPU fallback)")
    keys = keys.detach().contiguous()
    b, n_tok, c = keys.shape
    p = n_tok - tok0
    from .dino import TiledPlanes, split_tiled
    if isinstance(planes, str):
        if planes == "split":
            w3 = split_tiled(keys.view(b * n_tok, c)).blob
        elif planes == "fp32":
            w3 = None
        else:
            raise RuntimeError("mutual_nn_fused: unknown mode %r" % (planes,))
    else:
        if not (isinstance(planes, TiledPlanes) and (planes.rows, planes.cols) == (b * n_tok, c) and planes.blob.is_cuda):
            raise RuntimeError("mutual_nn_fused: planes must be the TiledPlanes [%d, %d] split of keys" % (b * n_tok, c))
        w3 = planes.blob
    si = src_img.to(device=keys.device, dtype=torch.int32).contiguous()
    ti = tgt_img.to(device=keys.device, dtype=torch.int32).contiguous()
    n = si.numel()
    md = None if mask_down is None else mask_down.detach().float().contiguous()
    if md is not None and tuple(md.shape) != (b, p):
        raise RuntimeError("mutual_nn_fused: mask must be [%d, %d]" % (b, p))
    fw = torch.empty(n, p, dtype=torch.int64, device=keys.device)
    bw = torch.empty(n, p, dtype=torch.int64, device=keys.device)
    nbytes = L.scp_mutual_nn_fused_workspace(n, p)
    ws = torch.empty(nbytes // 8, dtype=torch.int64, device=keys.device)
    capi.check(L.scp_mutual_nn_fused(capi.dev_ptr(keys, "keys"), ctypes.c_void_p(0 if w3 is None else w3.data_ptr()), b, n_tok, tok0, c,
                                     ctypes.c_void_p(si.data_ptr()), ctypes.c_void_p(ti.data_ptr()), capi.opt_ptr(md, "mask"), n, p,
                                     ctypes.c_void_p(fw.data_ptr()), ctypes.c_void_p(bw.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                     ctypes.c_size_t(nbytes), capi.current_stream()), "scp_mutual_nn_fused")
    return bw, fw


def rows_forward(scores, weights, tau):
    n, p, q = scores.shape
    w = weights.shape[-1]
    out = torch.empty(n, p, w, dtype=torch.float32, device=scores.device)
    stats = torch.empty(n, p, 2, dtype=torch.float32, device=scores.device)
    code = capi.lib().scp_softmax_rows_weighted_forward(
        capi.dev_ptr(scores, "scores"), capi.dev_ptr(weights, "weights"), w, float(tau), n, p, q,
        capi.dev_ptr(out, "out"), capi.dev_ptr(stats, "rowstats"), capi.current_stream())
    capi.check(code, "scp_softmax_rows_weighted_forward")
    return out, stats


def dual_backward(scores, rowmask, colmask, g_in, col=None, row=None):
    """col = (stats, out, g_out, grid, tau) or None; row = (stats, out, g_out, weights, tau) or None"""
    n, p, q = scores.shape
    g = torch.empty_like(scores)
    cs, co, gco, grid, tau_c = col if col is not None else (None, None, None, None, 0.)
    rs, ro, gro, wts, tau_r = row if row is not None else (None, None, None, None, 0.)
    code = capi.lib().scp_dual_softmax_backward(
        capi.dev_ptr(scores, "scores"), capi.opt_ptr(rowmask, "rowmask"), capi.opt_ptr(colmask, "colmask"),
        capi.opt_ptr(g_in, "g_scores_in"), capi.dev_ptr(g, "g_scores_out"),
        capi.opt_ptr(cs, "colstats"), capi.opt_ptr(co, "col_out"), capi.opt_ptr(gco, "g_col_out"),
        capi.opt_ptr(grid, "grid"), int(grid is not None and grid.dim() == 3), float(tau_c),
        capi.opt_ptr(rs, "rowstats"), capi.opt_ptr(ro, "row_out"), capi.opt_ptr(gro, "g_row_out"),
        capi.opt_ptr(wts, "weights"), 3 if wts is None else wts.shape[-1], float(tau_r), n, p, q,
        capi.current_stream())
    capi.check(code, "scp_dual_softmax_backward")
    return g


def _c(t):
    return None if t is None else t.contiguous().float()


class FeatureVertexMatch(Function):
    """pointcorr [B,P,V], match [B,P,3], imatch [B,2,V] from unit features (correspondence.py:42-53)"""

    @staticmethod
    def forward(ctx, img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh):
        img_feat, mesh_feat = img_feat.contiguous(), mesh_feat.contiguous()
        mask_down, verts, grid = _c(mask_down), _c(verts), _c(grid)
        pc = torch.bmm(img_feat.transpose(1, 2), mesh_feat.transpose(1, 2))       # [B,P,V], library GEMM (K = C)
        imatch, cstats = cols_forward(pc, mask_down, None, grid, tau_mesh, masked_out=pc)  # masks in place
        match, rstats = rows_forward(pc, verts, tau_img)
        ctx.save_for_backward(img_feat, mesh_feat, mask_down, verts, grid, pc, imatch, cstats, match, rstats)
        ctx.taus = (float(tau_img), float(tau_mesh))
        return pc, match, imatch

    @staticmethod
    def backward(ctx, g_pc, g_match, g_imatch):
        img_feat, mesh_feat, mask_down, verts, grid, pc, imatch, cstats, match, rstats = ctx.saved_tensors
        tau_img, tau_mesh = ctx.taus
        col = (cstats, imatch, _c(g_imatch), grid, tau_mesh) if g_imatch is not None else None
        row = (rstats, match, _c(g_match), verts, tau_img) if g_match is not None else None
        ds = dual_backward(pc, mask_down, None, _c(g_pc), col, row)                 # [B,P,V]
        dst = ds.transpose(1, 2)
        g_img = torch.bmm(mesh_feat.transpose(1, 2), dst) if ctx.needs_input_grad[0] else None      # [B,C,P]
        g_mesh = torch.bmm(dst, img_feat.transpose(1, 2)) if ctx.needs_input_grad[1] else None      # [B,V,C]
        return g_img, g_mesh, None, None, None, None, None


def fvm_fusable(img_feat, mesh_feat, hf, wf):
    """the fused kernels (csrc/corr_fused.hip) cover the shipped Wild6D configurations: 64 feature channels, 64x64 map"""
    return img_feat.is_cuda and img_feat.shape[1] == 64 and wf == 64 and hf % 2 == 0 and hf >= 2 and img_feat.shape[2] == hf * wf


class PooledScores:
    """what Correspondence.match hands to PretrainedCorrespondence.compute_cycle_loss as `pointcorr` in training: the 2x2
    pooled masked scores [B, hf*wf/4, V] (pretrained_corr.py:120-123 pools them first thing); the full [B, hf*wf, V] tensor is
    never formed"""

    def __init__(self, pooled, hf, wf, bridge=None):
        self.pooled, self.hf, self.wf = pooled, hf, wf
        # (grid_half [2,P/4], tau_mesh, xy [B,2,V], colstat [B,2,V]) when the fused forward also produced the column soft-argmax of
        # the pooled scores that pretrained_corr.py:123-126 takes (ops.vertex_bridge_match then skips its pass over `pooled`)
        self.bridge = bridge
        self.shape = (pooled.shape[0], hf * wf, pooled.shape[2])
        self.device, self.dtype = pooled.device, pooled.dtype


class FeatureVertexMatchFused(Function):
    """(pooled scores [B,P/4,V], match [B,P,3], imatch [B,2,V]) from unit features; scores live in registers only"""

    @staticmethod
    def forward(ctx, img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh, hf, wf, grid_half=None):
        img_feat, mesh_feat = img_feat.contiguous().float(), mesh_feat.contiguous().float()
        mask_down, verts, grid, grid_half = _c(mask_down), _c(verts), _c(grid), _c(grid_half)
        b, c, p = img_feat.shape
        v = mesh_feat.shape[1]
        dev = img_feat.device
        L = capi.lib()
        pooled = torch.empty(b, p // 4, v, dtype=torch.float32, device=dev)
        match = torch.empty(b, p, 3, dtype=torch.float32, device=dev)
        imatch = torch.empty(b, 2, v, dtype=torch.float32, device=dev)
        rowstat = torch.empty(b, p, 2, dtype=torch.float32, device=dev)
        colstat = torch.empty(b, v, 2, dtype=torch.float32, device=dev)
        bridge_xy = bridge_stat = None
        if grid_half is not None:
            if tuple(grid_half.shape) != (2, p // 4):
                raise RuntimeError("feature_vertex_match: grid_half must be [2, %d]" % (p // 4))
            bridge_xy = torch.empty(b, 2, v, dtype=torch.float32, device=dev)
            bridge_stat = torch.empty(b, v, 2, dtype=torch.float32, device=dev)
        nbytes = L.scp_fvm_workspace(b, hf, v)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        capi.check(L.scp_fvm_forward(capi.dev_ptr(img_feat, "img_feat"), capi.dev_ptr(mesh_feat, "mesh_feat"),
                                     capi.dev_ptr(mask_down, "mask_down"), capi.dev_ptr(verts, "verts"), capi.dev_ptr(grid, "grid"),
                                     float(tau_img), float(tau_mesh), b, c, hf, wf, v, capi.dev_ptr(pooled, "pooled"),
                                     capi.dev_ptr(match, "match"), capi.dev_ptr(imatch, "imatch"), capi.dev_ptr(rowstat, "rowstat"),
                                     capi.dev_ptr(colstat, "colstat"), capi.opt_ptr(grid_half, "grid_half"),
                                     capi.opt_ptr(bridge_xy, "bridge_xy"), capi.opt_ptr(bridge_stat, "bridge_colstat"),
                                     capi.dev_ptr(ws, "workspace"), ctypes.c_size_t(nbytes),
                                     capi.current_stream()), "scp_fvm_forward")
        ctx.save_for_backward(img_feat, mesh_feat, mask_down, verts, grid, match, imatch, rowstat, colstat)
        ctx.cfg = (float(tau_img), float(tau_mesh), hf, wf)
        if grid_half is None:
            return pooled, match, imatch
        # the bridge outputs are constants of this node: their gradient reaches `pooled` through BridgeInputs
        bridge_stat = bridge_stat.transpose(1, 2).contiguous()          # [B,2,V], the layout scp_dual_softmax_backward reads
        ctx.mark_non_differentiable(bridge_xy, bridge_stat)
        return pooled, match, imatch, bridge_xy, bridge_stat

    @staticmethod
    def backward(ctx, g_pooled, g_match, g_imatch, *_unused):
        img_feat, mesh_feat, mask_down, verts, grid, match, imatch, rowstat, colstat = ctx.saved_tensors
        tau_img, tau_mesh, hf, wf = ctx.cfg
        b, c, p = img_feat.shape
        v = mesh_feat.shape[1]
        g_img = torch.empty_like(img_feat) if ctx.needs_input_grad[0] else None
        g_mesh = torch.empty_like(mesh_feat) if ctx.needs_input_grad[1] else None
        capi.check(capi.lib().scp_fvm_backward(
            capi.dev_ptr(img_feat, "img_feat"), capi.dev_ptr(mesh_feat, "mesh_feat"), capi.dev_ptr(mask_down, "mask_down"),
            capi.dev_ptr(verts, "verts"), capi.dev_ptr(grid, "grid"), tau_img, tau_mesh, b, c, hf, wf, v,
            capi.dev_ptr(match, "match"), capi.dev_ptr(imatch, "imatch"), capi.dev_ptr(rowstat, "rowstat"),
            capi.dev_ptr(colstat, "colstat"), capi.opt_ptr(_c(g_match), "g_match"), capi.opt_ptr(_c(g_imatch), "g_imatch"),
            capi.opt_ptr(_c(g_pooled), "g_pooled"), capi.opt_ptr(g_img, "g_img"), capi.opt_ptr(g_mesh, "g_mesh"),
            capi.current_stream()), "scp_fvm_backward")
        return g_img, g_mesh, None, None, None, None, None, None, None, None


def pp_fusable(src_feat, tgt_feat):
    """the fused pixel <-> pixel kernels (csrc/corr_pp.hip): 64 channels, pixel counts that are multiples of 32"""
    return (src_feat.is_cuda and src_feat.dim() == 3 and tgt_feat.dim() == 3 and src_feat.shape[1] == 64 and tgt_feat.shape[1] == 64
            and src_feat.shape[2] % 32 == 0 and tgt_feat.shape[2] % 32 == 0)


class PixelPixelSoftArgmax(Function):
    """out [N,2,Q] = grid @ softmax_P(tau * masked(src^T tgt)) (correspondence.py:105-110); the [N,P,Q] scores live in registers"""

    @staticmethod
    def forward(ctx, src_feat, tgt_feat, src_mask, tgt_mask, grid, tau):
        src_feat, tgt_feat = src_feat.contiguous().float(), tgt_feat.contiguous().float()
        src_mask, tgt_mask, grid = _c(src_mask), _c(tgt_mask), _c(grid)
        n, c, p = src_feat.shape
        q = tgt_feat.shape[2]
        batched = 1 if grid.dim() == 3 else 0
        if grid.shape[-2:] != (2, p) or (batched and grid.shape[0] != n):
            raise RuntimeError("pixel_pixel_softargmax: grid must be [2,%d] or [%d,2,%d]" % (p, n, p))
        out = torch.empty(n, 2, q, dtype=torch.float32, device=src_feat.device)
        stats = torch.empty(n, 2, q, dtype=torch.float32, device=src_feat.device)
        capi.check(capi.lib().scp_pp_softargmax_forward(
            capi.dev_ptr(src_feat, "src_feat"), capi.dev_ptr(tgt_feat, "tgt_feat"), capi.opt_ptr(src_mask, "src_mask"),
            capi.opt_ptr(tgt_mask, "tgt_mask"), capi.dev_ptr(grid, "grid"), batched, float(tau), n, c, p, q,
            capi.dev_ptr(out, "out"), capi.dev_ptr(stats, "stats"), capi.current_stream()), "scp_pp_softargmax_forward")
        ctx.save_for_backward(src_feat, tgt_feat, src_mask, tgt_mask, grid, out, stats)
        ctx.cfg = (float(tau), batched)
        return out

    @staticmethod
    def backward(ctx, g_out):
        src_feat, tgt_feat, src_mask, tgt_mask, grid, out, stats = ctx.saved_tensors
        tau, batched = ctx.cfg
        n, c, p = src_feat.shape
        q = tgt_feat.shape[2]
        g_src = torch.empty_like(src_feat) if ctx.needs_input_grad[0] else None
        g_tgt = torch.empty_like(tgt_feat) if ctx.needs_input_grad[1] else None
        capi.check(capi.lib().scp_pp_softargmax_backward(
            capi.dev_ptr(src_feat, "src_feat"), capi.dev_ptr(tgt_feat, "tgt_feat"), capi.opt_ptr(src_mask, "src_mask"),
            capi.opt_ptr(tgt_mask, "tgt_mask"), capi.dev_ptr(grid, "grid"), batched, tau, n, c, p, q, capi.dev_ptr(out, "out"),
            capi.dev_ptr(stats, "stats"), capi.dev_ptr(_c(g_out), "g_out"), capi.opt_ptr(g_src, "g_src"), capi.opt_ptr(g_tgt, "g_tgt"),
            capi.current_stream()), "scp_pp_softargmax_backward")
        return g_src, g_tgt, None, None, None, None


class BridgeInputs(Function):
    """The two things ops.vertex_bridge_match reads from the pooled scores [B,P4,V], as ONE autograd node: the K selected rows of every
    pair (`rows` [N*K] indices into the [B*P4] rows) and the per-vertex column soft-argmax (already computed: out / stats [B,2,V] from
    FeatureVertexMatchFused).  Backward: the column part writes the whole gradient tensor once (scp_dual_softmax_backward), the
    gathered rows' gradients are added into it in place -- separately these were a zero-fill + scatter, a full write and autograd's
    sum of the two (5 passes over 84 MB instead of 2)."""

    @staticmethod
    def forward(ctx, scores, rows, grid, tau, out, stats):
        ctx.save_for_backward(scores, rows, _c(grid), out, stats)
        ctx.tau = float(tau)
        return scores.reshape(-1, scores.shape[-1]).index_select(0, rows), out.clone()

    @staticmethod
    def backward(ctx, g_sel, g_out):
        scores, rows, grid, out, stats = ctx.saved_tensors
        if g_out is not None:
            g = dual_backward(scores, None, None, None, (stats, out, _c(g_out), grid, ctx.tau), None)
        else:
            g = torch.zeros_like(scores)
        if g_sel is not None:
            g.view(-1, g.shape[-1]).index_add_(0, rows, g_sel)
        return g, None, None, None, None, None


class ColsSoftArgmax(Function):
    """out [N,2,Q] = grid @ softmax_P(tau * masked(scores))"""

    @staticmethod
    def forward(ctx, scores, rowmask, colmask, grid, tau):
        scores, rowmask, colmask, grid = scores.contiguous(), _c(rowmask), _c(colmask), _c(grid)
        out, stats = cols_forward(scores, rowmask, colmask, grid, tau)
        ctx.save_for_backward(scores, rowmask, colmask, grid, out, stats)
        ctx.tau = float(tau)
        return out

    @staticmethod
    def backward(ctx, g_out):
        scores, rowmask, colmask, grid, out, stats = ctx.saved_tensors
        g = dual_backward(scores, rowmask, colmask, None, (stats, out, _c(g_out), grid, ctx.tau), None)
        return g, None, None, None, None
