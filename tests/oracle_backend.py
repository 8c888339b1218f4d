"""tests/oracle_backend.py -- lets CPU tests drive the product's HOST logic (autograd glue,
SoftRenderer, model code) with the CPU oracle standing in for the HIP kernels.  Test-only: the
product never imports this; tests install it with pytest's monkeypatch."""
import numpy as np
import torch

from oracle import softras as oracle


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


def install(monkeypatch):
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    monkeypatch.setattr(native, "forward_soft_rasterize", forward_soft_rasterize)
    monkeypatch.setattr(native, "backward_soft_rasterize", backward_soft_rasterize)
