"""GPU parity of the HIP rasteriser (csrc/softras.hip through the C ABI) -- run with -m gpu.

Comparators: (1) the committed golden vectors recorded from the reference kernels, (2) the CPU
oracle on fresh seeded scenes, (3) size-independent properties at BASELINE.json's full size
(B=32, 256^2, 642 verts / 1280 faces).

Tolerances (north_star: 1e-4 relative fp32).  The HIP kernels keep the reference's evaluation order
and fp64 promotions and are built with -ffp-contract=off, so forward differences can only come
from the last ulp of expf (ocml vs glibc):
  forward images / aggregates : |d| <= 2e-6 + 1e-5 |ref| on EVERY pixel (no pixel may flip)
  faces_info                  : bit exact (no transcendental involved)
  gradients                   : summation order differs (wavefront tree + atomics vs raster order):
                                |d| <= 1e-4 * max|ref| element-wise and relative L2 <= 2e-5
"""
import numpy as np
import pytest
import torch

import golden_io
import scenes
from oracle import softras as oracle

pytestmark = pytest.mark.gpu

DEV = "cuda"


def hip_render(fv, ftex, grad=None, **kw):
    """product path: autograd Function -> native boundary -> HIP kernels"""
    from scp_amd.soft_renderer import functional as srf
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    assert "scp_amd" in native.__name__
    captured = {}
    orig = native.forward_soft_rasterize

    def spy(*a):
        out = orig(*a)
        captured["faces_info"], captured["aggrs_info"] = out[0], out[1]
        return out

    native.forward_soft_rasterize = spy
    try:
        fv_t = torch.tensor(fv, device=DEV, requires_grad=True)
        tex_t = torch.tensor(ftex, device=DEV, requires_grad=True)
        img = srf.soft_rasterize(fv_t, tex_t, **kw)
    finally:
        native.forward_soft_rasterize = orig
    out = {"soft_colors": img.detach().cpu().numpy(),
           "faces_info": captured["faces_info"].cpu().numpy(),
           "aggrs_info": captured["aggrs_info"].cpu().numpy()}
    if grad is not None:
        img.backward(torch.tensor(grad, device=DEV))
        out["grad_faces"] = fv_t.grad.cpu().numpy().reshape(fv.shape[0], -1, 9)
        out["grad_textures"] = tex_t.grad.cpu().numpy()
    return out


def assert_forward_close(got, ref):
    np.testing.assert_array_equal(got["faces_info"], ref["faces_info"])
    for key in ("soft_colors", "aggrs_info"):
        d = np.abs(got[key].astype(np.float64) - ref[key])
        tol = 2e-6 + 1e-5 * np.abs(ref[key])
        bad = d > tol
        assert not bad.any(), "%s: %d px out of tolerance, max abs diff %.3e" % (key, bad.sum(), d.max())


def assert_grad_close(got, ref, key):
    g, r = got[key].astype(np.float64).ravel(), ref[key].astype(np.float64).ravel()
    scale = np.abs(r).max()
    if scale == 0:
        assert np.abs(g).max() == 0
        return
    assert np.abs(g - r).max() <= 1e-4 * scale, "%s max abs diff %.3e vs scale %.3e" % (key, np.abs(g - r).max(), scale)
    assert np.linalg.norm(g - r) <= 2e-5 * np.linalg.norm(r), "%s rel L2 %.3e" % (key, np.linalg.norm(g - r) / np.linalg.norm(r))


@pytest.mark.parametrize("case", golden_io.softras_cases())
def test_hip_matches_reference_golden(case):
    d = golden_io.load(case)
    got = hip_render(d["face_vertices"], d["face_textures"], d["grad_soft_colors"], **golden_io.softras_kwargs(d))
    ref = dict(d)
    ref["grad_textures"] = d["grad_textures"].reshape(got["grad_textures"].shape)
    assert_forward_close(got, ref)
    assert_grad_close(got, ref, "grad_faces")
    assert_grad_close(got, ref, "grad_textures")


PASSES = {
    "mask": dict(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="hard", background_color=[0, 0, 0], texture_type="surface"),
    "depth": dict(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", background_color=[1, 1, 1], texture_type="vertex"),
    "softtex": dict(sigma_val=1e-3, gamma_val=1e-2, aggr_func_rgb="softmax", background_color=[1, 1, 1], texture_type="vertex"),
    "hardtex": dict(sigma_val=1e-4, gamma_val=1e-3, aggr_func_rgb="hard", background_color=[0, 0, 0], texture_type="vertex"),
}


@pytest.mark.parametrize("pname", list(PASSES))
@pytest.mark.parametrize("size,subdiv,n", [(96, 2, 3), (250, 3, 2)])
def test_hip_matches_oracle_fresh_scenes(pname, size, subdiv, n):
    """sizes that are not multiples of the 16-px tile, other meshes, other seeds"""
    v, f = scenes.bottle_like(subdiv)
    texkind = {"mask": "rand", "depth": "depth", "softtex": "rand", "hardtex": "canon"}[pname]
    fv, ftex = scenes.raster_inputs(v, f, n, seed=size + subdiv, tex=texkind)
    if pname == "mask":
        ftex = np.ones((n, f.shape[0], 1, 3), np.float32)
    grad = np.random.default_rng(size).standard_normal((n, 4, size, size)).astype(np.float32)
    kw = dict(image_size=size, dist_func="euclidean", aggr_func_alpha="prod", **PASSES[pname])
    ref = oracle.render(fv, ftex, grad_soft_colors=grad, **kw)
    got = hip_render(fv, ftex, grad, **kw)
    assert_forward_close(got, ref)
    assert_grad_close(got, ref, "grad_faces")
    assert_grad_close(got, ref, "grad_textures")


def test_high_res_dense_mesh_against_oracle():
    """BASELINE configs[4] geometry: 512x512, icosphere-4 topology (2562 verts / 5120 faces), one image,
    the sigma=1e-3 texture pass forward + backward (5120 faces > LIST_CAP exercises the chunked binning)"""
    v, f = scenes.bottle_like(4)
    assert v.shape[0] == 2562 and f.shape[0] == 5120
    fv, ftex = scenes.raster_inputs(v, f, 1, seed=77, tex="rand")
    grad = np.random.default_rng(5).standard_normal((1, 4, 512, 512)).astype(np.float32)
    kw = dict(image_size=512, dist_func="euclidean", aggr_func_alpha="prod", **PASSES["softtex"])
    ref = oracle.render(fv, ftex, grad_soft_colors=grad, **kw)
    got = hip_render(fv, ftex, grad, **kw)
    assert_forward_close(got, ref)
    assert_grad_close(got, ref, "grad_faces")
    assert_grad_close(got, ref, "grad_textures")


def test_ragged_and_empty_inputs():
    from scp_amd.soft_renderer import functional as srf
    # a single off-screen triangle: background everywhere, alpha exactly 0, zero gradients
    fv = torch.tensor([[[[5., 5., 3.], [5.1, 5., 3.], [5., 5.1, 3.]]]], device=DEV, requires_grad=True)
    tex = torch.ones(1, 1, 3, 3, device=DEV, requires_grad=True)
    img = srf.soft_rasterize(fv, tex, 40, [0.25, 0.5, 0.75], sigma_val=1e-4, texture_type="vertex")
    assert torch.equal(img[0, 3], torch.zeros(40, 40, device=DEV))
    assert torch.allclose(img[0, :3].mean((1, 2)).cpu(), torch.tensor([0.25, 0.5, 0.75]))
    img.sum().backward()
    assert fv.grad.abs().max().item() == 0
    # more faces than one binning list holds (LIST_CAP) all covering one tile, degenerate faces too
    rng = np.random.default_rng(3)
    n_f = 3000
    tri = rng.uniform(-0.05, 0.05, (1, n_f, 3, 3)).astype(np.float32)
    tri[..., 2] = rng.uniform(3, 6, (1, n_f, 3))
    tri[0, 5] = tri[0, 5, :1]          # zero-area face
    tri[0, 7, :, :2] = 0.0             # all corners identical in x,y
    t = rng.uniform(0, 1, (1, n_f, 3, 3)).astype(np.float32)
    grad = rng.standard_normal((1, 4, 48, 48)).astype(np.float32)
    kw = dict(image_size=48, sigma_val=1e-4, gamma_val=1e-2, aggr_func_rgb="softmax", texture_type="vertex",
              background_color=[1, 1, 1], dist_func="euclidean", aggr_func_alpha="prod")
    ref = oracle.render(tri, t, grad_soft_colors=grad, **kw)
    got = hip_render(tri, t, grad, **kw)
    assert_forward_close(got, ref)
    assert_grad_close(got, ref, "grad_faces")


def test_native_boundary_rejects_bad_tensors():
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    fv = torch.rand(1, 2, 9)
    with pytest.raises(RuntimeError):
        native.forward_soft_rasterize(fv, fv, fv, fv, fv, 16, 1., 100., 1e-3, 1e-4, 2, 9.2, 1e-4, 1, 2, 1, True)
    g = torch.rand(1, 2, 18, device=DEV)[:, :, ::2]
    with pytest.raises(RuntimeError):
        native.forward_soft_rasterize(g, g, g, g, g, 16, 1., 100., 1e-3, 1e-4, 2, 9.2, 1e-4, 1, 2, 1, True)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full size: B=32, 256x256, 642 verts / 1280 faces
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_scene():
    v, f = scenes.bottle_like(3)
    assert v.shape[0] == 642 and f.shape[0] == 1280
    fv, ftex = scenes.raster_inputs(v, f, 32, seed=2024, tex="rand")
    return fv, ftex


def test_full_size_subset_against_oracle(full_scene):
    fv, ftex = full_scene
    grad = np.random.default_rng(1).standard_normal((32, 4, 256, 256)).astype(np.float32)
    kw = dict(image_size=256, dist_func="euclidean", aggr_func_alpha="prod", **PASSES["softtex"])
    got = hip_render(fv, ftex, grad, **kw)
    pick = [0, 13, 31]
    ref = oracle.render(fv[pick], ftex[pick], grad_soft_colors=grad[pick], **kw)
    sub = {k: v[pick] for k, v in got.items()}
    assert_forward_close(sub, ref)
    assert_grad_close(sub, ref, "grad_faces")
    assert_grad_close(sub, ref, "grad_textures")


def test_full_size_properties(full_scene):
    from scp_amd.soft_renderer import functional as srf
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    fv, ftex = full_scene
    fv_t, tex_t = torch.tensor(fv, device=DEV), torch.tensor(ftex, device=DEV)
    kw_d = dict(image_size=256, dist_func="euclidean", aggr_func_alpha="prod", **PASSES["depth"])
    kw_m = dict(image_size=256, dist_func="euclidean", aggr_func_alpha="prod", **PASSES["mask"])
    depth = srf.soft_rasterize(fv_t, tex_t, **kw_d)
    mask = srf.soft_rasterize(fv_t, torch.ones(32, 1280, 1, 3, device=DEV), **kw_m)
    # (1) alpha in [0,1]; the mask and depth passes share sigma -> identical alpha
    assert depth[:, 3].min() >= 0 and depth[:, 3].max() <= 1
    assert torch.equal(depth[:, 3], mask[:, 3])
    # (2) batch-permutation equivariance, bit exact (checks tile -> image mapping at full grid size)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0)).to(DEV)
    depth_p = srf.soft_rasterize(fv_t[perm], tex_t[perm], **kw_d)
    assert torch.equal(depth_p, depth[perm])
    # (3) run-to-run determinism of the forward
    assert torch.equal(srf.soft_rasterize(fv_t, tex_t, **kw_d), depth)
    # (4) backward is linear in the incoming gradient
    fv_g = fv_t.clone().requires_grad_(True)
    img = srf.soft_rasterize(fv_g, tex_t, **kw_d)
    g1 = torch.randn(img.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    g2 = torch.randn(img.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(2))
    ga, = torch.autograd.grad(img, fv_g, g1, retain_graph=True)
    gb, = torch.autograd.grad(img, fv_g, g2, retain_graph=True)
    gc, = torch.autograd.grad(img, fv_g, g1 + 2 * g2)
    assert torch.linalg.norm(gc - (ga + 2 * gb)) <= 1e-5 * torch.linalg.norm(gc)
    # (5) the instrumentation kernel agrees with the oracle's pair count on a few images
    n_gpu = native.count_pairs(fv_t[:3].reshape(3, 1280, 9).contiguous(), 256, 1e-4, float(np.log(1. / 1e-4 - 1.)))
    n_cpu = oracle.count_pairs(np.ascontiguousarray(fv[:3].reshape(3, 1280, 9)), 256, 1e-4, float(np.log(1. / 1e-4 - 1.)))
    assert n_gpu == n_cpu


def test_exact_division_by_hoisted_divisor_is_bit_identical_to_ieee_division():
    """csrc/softras.hip replaces a / b by q = a * RN(1/b) + two fused residual corrections wherever b is a face / pixel /
    pass constant; the coverage decisions and the knife-edge gamma = 1e-4 softmax need the IEEE quotient exactly: 2^31
    generated operand pairs (uniform + adversarial mantissas), zero mismatches allowed"""
    import ctypes
    from scp_amd import capi
    bad = torch.zeros(1, dtype=torch.int64, device=DEV)
    for seed in (1, 2):
        capi.check(capi.lib().scp_selftest_exact_division(ctypes.c_ulonglong(1 << 30), seed, ctypes.c_void_p(bad.data_ptr()),
                                                          capi.current_stream()), "selftest")
    torch.cuda.synchronize()
    assert int(bad.item()) == 0, "%d of 2^31 quotients differ from IEEE division" % int(bad.item())


def test_dual_forward_equals_two_separate_passes_bit_for_bit():
    """scp_soft_rasterize_forward_dual (depth pass + canonical-colour hard pass in one launch) against two
    scp_soft_rasterize_forward calls: both images, both aggregate buffers and the backward of the primary pass"""
    from scp_amd.soft_renderer import functional as srf
    v, f = scenes.bottle_like(3)
    fv, ftex = scenes.raster_inputs(v, f, 5, seed=31, tex="depth")
    _, fcanon = scenes.raster_inputs(v, f, 5, seed=31, tex="canon")
    for size in (256, 100):
        common = dict(image_size=size, dist_func="euclidean", aggr_func_alpha="prod")
        fv_a = torch.tensor(fv, device=DEV, requires_grad=True)
        tex_a = torch.tensor(ftex, device=DEV, requires_grad=True)
        canon = torch.tensor(fcanon, device=DEV)
        depth = srf.soft_rasterize(fv_a, tex_a, **common, **PASSES["depth"])
        hard = srf.soft_rasterize(fv_a.detach(), canon, **common, **PASSES["hardtex"])
        fv_b = torch.tensor(fv, device=DEV, requires_grad=True)
        tex_b = torch.tensor(ftex, device=DEV, requires_grad=True)
        depth2, hard2 = srf.soft_rasterize_dual(fv_b, tex_b, canon, size, PASSES["depth"]["background_color"],
                                                PASSES["hardtex"]["background_color"], sigma_val=1e-4, gamma_val=1e-4)
        assert torch.equal(depth, depth2) and torch.equal(hard, hard2)
        assert not hard2.requires_grad
        g = torch.randn(depth.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(size))
        depth.backward(g)
        depth2.backward(g)
        # same kernel, same inputs; only the unordered atomics differ
        assert torch.linalg.norm(fv_a.grad - fv_b.grad) <= 2e-5 * torch.linalg.norm(fv_a.grad)
        assert torch.linalg.norm(tex_a.grad - tex_b.grad) <= 2e-5 * torch.linalg.norm(tex_a.grad)


def test_render_all_fused_equals_unfused():
    """Renderer.render_all with the hardtex pass riding on the depth launch == the four-pass form, output by output"""
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.model import MeshNet
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=3, train=True)
    torch.manual_seed(0)
    model = MeshNet(opts, prior=scenes.bottle_like(3)).to(DEV)
    B, V = 3, model.mesh.num_verts
    g = torch.Generator().manual_seed(5)
    pred_v = (model.mesh.mean_v.detach().cpu()[None] + 0.01 * torch.randn(B, V, 3, generator=g)).to(DEV)
    tex = torch.rand(B, V, 3, generator=g).to(DEV)
    rot = torch.tensor(scenes.random_rotations(B, np.random.default_rng(1)), dtype=torch.float32, device=DEV)
    trans = torch.tensor([[[0.02, -0.03, 5.1]]], device=DEV).repeat(B, 1, 1)
    foc = torch.full((B, 2), 5.9, device=DEV)
    pp = torch.zeros(B, 2, device=DEV)
    faces = model.mesh.faces[None].expand(B, -1, -1)
    r = model.renderer
    r.share_hardtex_with_depth = True
    fused = r.render_all(pred_v, faces, tex, foc, pp, rot, trans, None)
    r.share_hardtex_with_depth = False
    plain = r.render_all(pred_v, faces, tex, foc, pp, rot, trans, None)
    r.share_hardtex_with_depth = True
    for a, b in zip(fused, plain):
        assert torch.equal(a, b)


@pytest.mark.parametrize("pname,size,subdiv,n", [("softtex", 250, 3, 2), ("depth", 96, 2, 3), ("softtex", 256, 3, 32), ("depth", 256, 3, 32),
                                                  ("softtex", 512, 4, 1)])
def test_pair_queue_forward_is_bit_identical_to_the_per_face_forward(pname, size, subdiv, n, monkeypatch):
    """round 6: the pair-queue forward (raster_forward_pq_kernel, opt-in SCP_RASTER_FWD=pq: coverage arithmetic per PAIR at full lane
    use, the order-dependent per-pixel update applied in face order afterwards; measured slower than the per-face kernel, so not the
    default).  Same expressions in the same order per pixel: images and aggregates must equal the per-face kernel's BIT FOR BIT -- ragged
    image sizes, the headline batch, the 5120-face mesh (chunked binning), and an alpha mode / distance mode other than the step's"""
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    v, f = scenes.bottle_like(subdiv)
    fv, ftex = scenes.raster_inputs(v, f, n, seed=3 * size + subdiv, tex="depth" if pname == "depth" else "rand")
    kw = dict(image_size=size, dist_func="euclidean", aggr_func_alpha="prod", **PASSES[pname])

    def render(**over):
        captured = {}
        orig = native.forward_soft_rasterize

        def spy(*a):
            out = orig(*a)
            captured["aggrs_info"] = out[1]
            return out
        native.forward_soft_rasterize = spy
        try:
            from scp_amd.soft_renderer import functional as srf
            img = srf.soft_rasterize(torch.tensor(fv, device=DEV), torch.tensor(ftex, device=DEV), **dict(kw, **over))
        finally:
            native.forward_soft_rasterize = orig
        return img.clone(), captured["aggrs_info"].clone()

    variants = [{}] if n > 3 else [{}, dict(aggr_func_alpha="sum"), dict(dist_func="barycentric"), dict(aggr_func_alpha="hard", dist_func="hard")]
    for over in variants:
        monkeypatch.delenv("SCP_RASTER_FWD", raising=False)
        img0, ag0 = render(**over)
        monkeypatch.setenv("SCP_RASTER_FWD", "pq")
        img1, ag1 = render(**over)
        monkeypatch.delenv("SCP_RASTER_FWD")
        as_bits = lambda t: t.view(torch.int32)
        assert torch.equal(as_bits(img0), as_bits(img1)), (over, float((img0 - img1).abs().max()), int((as_bits(img0) != as_bits(img1)).sum()))
        assert torch.equal(as_bits(ag0), as_bits(ag1)), (over, int((as_bits(ag0) != as_bits(ag1)).sum()))
    assert float(img1[:, 3].max()) > 0.5, "the scene must cover something"


def test_pair_queue_dual_forward_is_bit_identical_to_the_per_face_forward(monkeypatch):
    """the fused depth + hard-colour forward (scp_soft_rasterize_forward_dual, the step's sigma = 1e-4 group) on the pair queue vs the
    per-face kernel: both images and both aggregate buffers bit for bit, at the headline batch and on a ragged size"""
    import ctypes
    from scp_amd import capi
    for size, n in ((256, 32), (250, 2)):
        v, f = scenes.bottle_like(3)
        fv, ftex = scenes.raster_inputs(v, f, n, seed=size, tex="depth")
        _, ftex2 = scenes.raster_inputs(v, f, n, seed=size + 1, tex="canon")
        fv_t, tex_t, tex2_t = (torch.tensor(x, device=DEV).contiguous() for x in (fv, ftex, ftex2))
        nf = fv.shape[1]
        p = capi.RasterParams(batch_size=n, num_faces=nf, image_size=size, texture_size=3, near_=1.0, far_=100.0, eps=1e-3, sigma_val=1e-4,
                              dist_eps=9.21024036697585, gamma_val=1e-4, func_id_dist=2, func_id_rgb=1, func_id_alpha=2,
                              texture_sample_type=1, double_side=0)

        def run():
            info = torch.zeros(n, nf, 27, device=DEV)
            ag, ag2 = torch.zeros(n, 2, size, size, device=DEV), torch.zeros(n, 2, size, size, device=DEV)
            img, img2 = torch.ones(n, 4, size, size, device=DEV), torch.zeros(n, 4, size, size, device=DEV)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            capi.check(capi.lib().scp_soft_rasterize_forward_dual(P(fv_t.view(n, nf, 9)), P(tex_t.view(n, nf, 9)), P(info), P(ag), P(img),
                                                                   P(tex2_t.view(n, nf, 9)), P(ag2), P(img2), ctypes.byref(p), capi.current_stream()),
                       "forward_dual")
            return [t.view(torch.int32).clone() for t in (img, ag, img2, ag2)]
        monkeypatch.delenv("SCP_RASTER_FWD", raising=False)
        ref = run()
        monkeypatch.setenv("SCP_RASTER_FWD", "pq")
        got = run()
        monkeypatch.delenv("SCP_RASTER_FWD")
        for a, b, name in zip(ref, got, ("soft", "aggrs", "hard", "aggrs_hard")):
            assert torch.equal(a, b), (size, name, int((a != b).sum()))
        assert int((got[3][:, 1].view(torch.float32) >= 0).sum()) > 0, "some pixel must have a z-buffer winner"
