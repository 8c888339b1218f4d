"""oracle/backend.py -- TEST INFRASTRUCTURE, NOT PRODUCT.  Lets CPU tests (and the cpu_baseline leg of bench.py) drive the product's HOST logic (autograd glue,
SoftRenderer, model code) with the CPU oracle standing in for EVERY HIP kernel (rasteriser,
correspondence reductions, ViT attention / LayerNorm, nearest neighbour).  Test-only: the product never
imports this; tests install it with pytest's monkeypatch.  Without it the product raises on CPU tensors."""
import numpy as np
import torch

from oracle import corr as oracle_corr
from oracle import softras as oracle
from oracle import vit as oracle_vit


def _np(t):
    return t.detach().cpu().numpy()


def forward_soft_rasterize(faces, textures, faces_info, aggrs_info, soft_colors, *scal):
    fi, ai, sc = _np(faces_info), _np(aggrs_info), _np(soft_colors)
    oracle.forward(np.ascontiguousarray(_np(faces)), np.ascontiguousarray(_np(textures)), fi, ai, sc, *scal)
    for dst, src in ((faces_info, fi), (aggrs_info, ai), (soft_colors, sc)):
        if dst.data_ptr() != torch.from_numpy(src).data_ptr():
            dst.copy_(torch.from_numpy(src))
    return [faces_info, aggrs_info, soft_colors]


def backward_soft_rasterize(faces, textures, soft_colors, faces_info, aggrs_info, grad_faces,
                            grad_textures, grad_soft_colors, *scal):
    gf, gt = _np(grad_faces), _np(grad_textures)
    oracle.backward(_np(faces), _np(textures), _np(soft_colors), _np(faces_info), _np(aggrs_info), gf, gt,
                    np.ascontiguousarray(_np(grad_soft_colors)), *scal)
    for dst, src in ((grad_faces, gf), (grad_textures, gt)):
        if dst.data_ptr() != torch.from_numpy(src).data_ptr():
            dst.copy_(torch.from_numpy(src))
    return [grad_faces, grad_textures]


def feature_vertex_match(img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh):
    return oracle_corr.match_oracle(img_feat, mesh_feat, mask_down, verts, grid, tau_img, tau_mesh)


def fused_attention(qkv, b, n, heads, head_dim, scale):
    return oracle_vit.attention_oracle(qkv, heads, scale)


def add_layernorm(x, branch, norm):
    return oracle_vit.add_layernorm_oracle(x, branch, norm.weight, norm.bias, norm.eps)


def mutual_nn(src_feat, tgt_feat, src_mask, tgt_mask):
    bw, fw, _ = oracle_corr.mutual_nn_oracle(src_feat, tgt_feat, src_mask, tgt_mask)
    return bw, fw


def mutual_nn_pairs(keys, src_img, tgt_img, mask_down, tok0=1):
    """CPU stand-in of scp_mutual_nn_fused: gather the pairs' feature maps and run the reference formulation"""
    feat = keys[:, tok0:, :].transpose(1, 2)                      # n_images, C, P
    bw, fw, _ = oracle_corr.mutual_nn_oracle(feat[src_img], feat[tgt_img], mask_down[src_img], mask_down[tgt_img])
    return bw, fw


def install(monkeypatch):
    import scp_amd.dino as dino
    import scp_amd.mesh as mesh
    import scp_amd.ops as ops
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    monkeypatch.setattr(native, "forward_soft_rasterize", forward_soft_rasterize)
    monkeypatch.setattr(native, "backward_soft_rasterize", backward_soft_rasterize)
    monkeypatch.setattr(ops, "feature_vertex_match", feature_vertex_match)
    monkeypatch.setattr(ops, "cols_softargmax", oracle_corr.cols_softargmax_oracle)
    monkeypatch.setattr(ops, "mutual_nn", mutual_nn)
    monkeypatch.setattr(ops, "mutual_nn_pairs", mutual_nn_pairs)
    monkeypatch.setattr(dino, "fused_attention", fused_attention)
    monkeypatch.setattr(dino, "add_layernorm", add_layernorm)
    monkeypatch.setattr(mesh, "nearest_index", oracle_corr.nearest_index_oracle)
