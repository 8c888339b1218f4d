"""G2: Renderer.render_all by itself against the reference's recording (tests/golden/render_all_bottle_b2.npz, made by
make_golden.py gen_render from model/module/renderer.py:38-73): the nine outputs and the gradients of a fixed random functional
w.r.t. pred_v, tex, rotation, translation.  CPU: host logic + C oracle rasteriser; GPU: the HIP rasteriser (dual depth / canonical
pass, mask shared with the depth pass).  Per-pixel band of SURVEY F12 (>= 97 % of pixels within 1e-4; alpha: <= 0.1 % of the pixels off by > 5e-2, max-abs within the
spread the reference itself shows under 1e-7 perturbations of its inputs, recorded in the fixture);
image sums (the loss-like quantities) within 1e-4 relative; gradients cos >= 0.9999, norm within 1e-2."""
import numpy as np
import pytest
import torch

import golden_io

NAMES = ("mask_render", "tex_render", "depth_render", "match_gt", "imatch_gt", "tex_mask", "depth_mask", "match_mask", "depth_weight")


def _run(device):
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.model import MeshNet
    d = golden_io.load("render_all_bottle_b2")
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True)
    step = golden_io.load("step_laptopflags_bottle_b2x2")
    model = MeshNet(opts, prior=(step["prior_verts"], step["prior_faces"])).to(device)
    t = lambda k, grad=False: torch.tensor(d[k], device=device).requires_grad_(grad)
    pred_v, tex, rot, trans = t("pred_v", True), t("tex", True), t("rotation", True), t("translation", True)
    assert np.array_equal(model.mesh.faces.cpu().numpy(), d["faces"])
    faces = model.mesh.faces[None].expand(2, -1, -1)
    outs = model.renderer.render_all(pred_v, faces, tex, t("foc_crop"), t("pp_crop"), rot, trans, None)
    g = torch.Generator().manual_seed(int(d["weights_seed"]))
    torch.rand(2, pred_v.shape[1], 3, generator=g)                       # the generator's draw for `tex`, recorded in the fixture
    weights = {k: torch.randn(o.shape, generator=g).to(device) for k, o in zip(NAMES, outs)}
    # the functional's terms, as recorded: every differentiable output except the canonical-xyz pass (no backward: SURVEY F8)
    req = [str(k) for k in d["requires_grad"]]
    assert all(dict(zip(NAMES, outs))[k].requires_grad for k in req)
    sum((o * weights[k]).sum() for k, o in zip(NAMES, outs) if k in req).backward()
    return d, dict(zip(NAMES, outs)), {"pred_v": pred_v.grad, "tex": tex.grad, "rotation": rot.grad, "translation": trans.grad}


def _check(d, outs, grads):
    for k, o in outs.items():
        got = o.detach().double().cpu()
        ref_stats = d["stats_" + k]
        mine = np.array([got.sum().item(), got.abs().sum().item(), (got * got).sum().item()])
        # |sum|, sum of squares (loss-like): 1e-4, or what the REFERENCE's own render moves them by when its inputs are perturbed at
        # 1e-7 .. 1e-6 (cond_stats_*: a handful of pixels flipping across the hard footprint cut move the depth planes' sums by ~1e-4)
        tol = np.maximum(1e-4, 1.5 * d["cond_stats_" + k])
        assert (np.abs(mine[1:] - ref_stats[1:]) <= tol * np.abs(ref_stats[1:])).all(), (k, mine[1:], ref_stats[1:], tol)
        sub = got.numpy()[..., ::2, ::2] if got.dim() >= 3 and got.shape[-1] == 256 else got.numpy()
        ref = d["out_" + k]
        diff = np.abs(sub - ref)
        if k in ("imatch_gt", "depth_weight"):
            # per vertex: 1e-4, except where the reference's own value moves under 1e-7-level input perturbations (depth_weight samples
            # the depth image at the projected vertex: a silhouette vertex sees the pixels that flip) -- by at most that recorded move
            band = 1.5 * float(d["cond_maxabs_" + k].max())
            loose = diff > 1e-5 + 1e-4 * np.abs(ref)
            assert loose.mean() <= 0.02 and (diff[loose] <= band).all(), (k, loose.mean(), diff.max(), band)
        else:
            assert (diff <= 1e-4 + 1e-4 * np.abs(ref)).mean() >= 0.97, (k, (diff <= 1e-4 + 1e-4 * np.abs(ref)).mean())
            if k.endswith("mask") or k == "mask_render":
                # Per-pixel alpha is knife-edge at sigma = 1e-4: the REFERENCE's own render moves single pixels by up to 0.48 when its
                # geometry inputs are perturbed by 1e-7 (cond_maxabs_*, recorded by make_golden.py gen_render from the reference
                # itself), i.e. by what any backend's rounding of the projection does.  A max-abs bound on alpha therefore has to be
                # that recorded spread (the 5e-2 of SURVEY F12 was one observation, and the GPU side sat at 0.045 .. 0.075 around it
                # depending on the last bit of the projected vertices); what IS stable is how FEW pixels move: <= 0.1 % by > 5e-2.
                assert diff.max() <= max(5e-2, 1.5 * float(d["cond_maxabs_" + k].max())), (k, diff.max())
                assert (diff > 5e-2).mean() <= 1e-3, (k, (diff > 5e-2).mean())
                # ... and a localized regression (a few dozen pixels far off) must not hide under the recorded max: at most 1e-4 of
                # the pixels (~13 of 2 x 256 x 256) may differ by more than 0.2
                assert (diff > 0.2).mean() <= 1e-4, (k, (diff > 0.2).mean(), int((diff > 0.2).sum()))
    for k, gr in grads.items():
        a, b = gr.detach().double().cpu().numpy().ravel(), d["grad_" + k].astype(np.float64).ravel()
        cos = a @ b / (np.linalg.norm(a) * np.linalg.norm(b))
        rel = abs(np.linalg.norm(a) - np.linalg.norm(b)) / np.linalg.norm(b)
        print("grad %-12s cos %.7f norm rel %.2e" % (k, cos, rel))
        # SURVEY F12's band (cosine >= 0.9999, norm within 1e-2) -- or, where the reference's OWN gradient moves further when its
        # inputs are perturbed by 1e-7 .. 3e-7 (cond_grad_*: the silhouette terms), that recorded move with slack
        min_cos, max_rel = (float(x) for x in d["cond_grad_" + k])
        assert cos >= min(0.9999, 1.0 - 2.0 * (1.0 - min_cos)) and rel <= max(1e-2, 2.0 * max_rel), (k, cos, rel, min_cos, max_rel)


def test_render_all_matches_reference_cpu(monkeypatch):
    import oracle_backend
    oracle_backend.install(monkeypatch)
    _check(*_run("cpu"))


@pytest.mark.gpu
def test_render_all_matches_reference_gpu():
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    assert native.forward_soft_rasterize.__module__.startswith("scp_amd"), "HIP path must be the one that runs"
    _check(*_run("cuda"))
