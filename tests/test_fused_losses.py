"""The fused per-pixel loss ops (csrc/losses.hip through scp_amd/fused_losses.py) against the torch compositions of scp_amd/losses.py
evaluated in float64 (they restate model/util/loss_utils.py:236-252, :273-284, :317-320): values and gradients."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))

pytestmark = pytest.mark.gpu


def _inputs(b, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    def rnd(*s):
        return torch.rand(*s, generator=g)
    blob = (rnd(b, h, w) > 0.45).float()
    # soft alpha with exact zeros and ones (as the rasteriser leaves them) and a soft band
    alpha = (rnd(b, h, w) * 1.6 - 0.3).clamp(0, 1) * (rnd(b, h, w) > 0.2)
    depth_pred = rnd(b, h, w) * 2 + 0.5
    depth_out = torch.stack((rnd(b, h, w), rnd(b, h, w), depth_pred, alpha), 1)
    mask = blob
    depth = (rnd(b, h, w) * 3 + 0.2) * (rnd(b, h, w) > 0.15)
    depth[:, : h // 8] *= 3.0                       # some |diff| > 1 pixels: the relu clamp and its zero gradient
    match = rnd(b, 3, h, w) - 0.5
    match_out = torch.cat((rnd(b, 3, h, w) - 0.5, (rnd(b, 1, h, w) > 0.4).float()), 1)
    match[:, :, 0, :4] = match_out[:, :3, 0, :4]    # exact zero distance: gradient 0, not NaN
    tex_out = torch.cat((rnd(b, 3, h, w), (rnd(b, 1, h, w) * 1.4 - 0.2).clamp(0, 1)), 1)
    img = rnd(b, 3, h, w)
    return [t.to(device) for t in (depth_out, match, match_out, depth, mask, tex_out, img)]


def _reference(losses, depth_out, match, match_out, depth, mask, tex_out, img):
    mask_l = losses.compute_mask_loss(None, mask, depth_out[:, 3])
    depth_l, _ = losses.compute_depth_loss(depth, depth_out[:, 2], depth_out[:, 3], mask)
    match_l = losses.compute_match_loss(match, match_out[:, :3], match_out[:, 3], mask)
    tex_l = losses.compute_texture_loss(img, mask, tex_out[:, :3], tex_out[:, 3])
    return mask_l, depth_l, match_l, tex_l


@pytest.mark.parametrize("b,h,w", [(3, 24, 64), (2, 16, 256), (2, 8, 32)])
def test_values_and_gradients(b, h, w):
    from scp_amd import fused_losses, losses
    dev = torch.device("cuda:0")
    depth_out, match, match_out, depth, mask, tex_out, img = _inputs(b, h, w, 5 + w, dev)
    wts = [torch.rand(b, generator=torch.Generator().manual_seed(k)).to(dev) + 0.5 for k in range(4)]

    a = [t.clone().requires_grad_(True) for t in (depth_out, match, tex_out)]
    got = list(fused_losses.depth_group_losses(a[0], match_out, a[1], depth, mask)) + [fused_losses.texture_loss(a[2], img, mask)]
    sum((g * wt).sum() for g, wt in zip(got, wts)).backward()

    r = [t.double().clone().requires_grad_(True) for t in (depth_out, match, tex_out)]
    ref = _reference(losses, r[0], r[1], match_out.double(), depth.double(), mask.double(), r[2], img.double())
    sum((g * wt.double()).sum() for g, wt in zip(ref, wts)).backward()

    for name, g, e in zip(("mask", "depth", "match", "texture"), got, ref):
        assert torch.allclose(g.double(), e, rtol=2e-5, atol=1e-7), (name, g, e)
    for name, x, y in zip(("depth_out", "match", "tex_out"), a, r):
        err = (x.grad.double() - y.grad).abs().max().item()
        scale = y.grad.abs().max().item()
        assert err <= 2e-5 * scale + 1e-9, (name, err, scale)
    assert torch.isfinite(a[1].grad).all()
    assert (a[0].grad[:, :2] == 0).all()


def test_against_fp32_composition_in_the_step_shape():
    """same numbers as the fp32 torch composition the step used before, at the step's image size"""
    from scp_amd import fused_losses, losses
    dev = torch.device("cuda:0")
    depth_out, match, match_out, depth, mask, tex_out, img = _inputs(4, 256, 256, 11, dev)
    got = list(fused_losses.depth_group_losses(depth_out, match_out, match, depth, mask)) + [fused_losses.texture_loss(tex_out, img, mask)]
    ref = _reference(losses, depth_out, match, match_out, depth, mask, tex_out, img)
    for g, e in zip(got, ref):
        assert torch.allclose(g, e, rtol=1e-5, atol=1e-7)


def test_rejects_unsupported_width():
    from scp_amd import fused_losses
    dev = torch.device("cuda:0")
    depth_out, match, match_out, depth, mask, tex_out, img = _inputs(1, 8, 48, 3, dev)
    assert not fused_losses.covers(img, mask)
    with pytest.raises(RuntimeError):
        fused_losses.texture_loss(tex_out, img, mask)
