"""Host-side SoftRas glue (scp_amd.soft_renderer: autograd Function, Mesh, SoftRenderer) checked
on CPU against the golden vectors, with the oracle substituted for the HIP kernels by monkeypatch."""
import numpy as np
import pytest
import torch

import golden_io
import oracle_backend


@pytest.mark.parametrize("case", ["softras_octa64_depth", "softras_octa64_mask", "softras_bottle128_softtex",
                                  "softras_octa64_surface_r2", "softras_octa64_sum_alpha"])
def test_autograd_glue_reproduces_golden(case, monkeypatch):
    oracle_backend.install(monkeypatch)
    from scp_amd.soft_renderer import functional as srf
    d = golden_io.load(case)
    fv = torch.tensor(d["face_vertices"], requires_grad=True)
    tex = torch.tensor(d["face_textures"], requires_grad=True)
    img = srf.soft_rasterize(fv, tex, **golden_io.softras_kwargs(d))
    np.testing.assert_array_equal(img.detach().numpy(), d["soft_colors"])
    img.backward(torch.tensor(d["grad_soft_colors"]))
    np.testing.assert_array_equal(fv.grad.numpy().reshape(d["grad_faces"].shape), d["grad_faces"])
    np.testing.assert_array_equal(tex.grad.numpy(), d["grad_textures"].reshape(tex.shape))


def test_mesh_gathers_and_shapes():
    import scp_amd.soft_renderer as sr
    v = torch.rand(2, 5, 3)
    f = torch.tensor([[0, 1, 2], [2, 3, 4]])[None].repeat(2, 1, 1)
    m = sr.Mesh(v, f, torch.rand(2, 5, 3), texture_type="vertex")
    assert m.face_vertices.shape == (2, 2, 3, 3) and m.face_textures.shape == (2, 2, 3, 3)
    assert torch.equal(m.face_vertices[1, 1, 2], v[1, 4])
    m2 = sr.Mesh(v, f)
    assert m2.textures.shape == (2, 2, 1, 3) and m2.texture_type == "surface"
