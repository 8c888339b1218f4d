"""colour jitter + normalize: torch composition (host mirror) vs the numpy oracle on CPU; the fused HIP
pass vs the oracle on the GPU.  torchvision is absent, so both are pinned to the published formulas only
(oracle/imgops.py header: parity unpinned)."""
import itertools

import numpy as np
import pytest
import torch

from oracle import imgops as oracle_imgops
from scp_amd import imgops

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _images(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, 3, h, w, generator=g)
    img[0, :, : h // 4] = 0.5            # grey patch: max == min branch of rgb->hsv
    img[-1, :, :, : w // 4] = 0.0        # black
    img[-1, 0, h // 2:, w // 2:] = 1.0   # saturated red block
    return img


CASES = [([0, 1, 2, 3], [1.13, 0.84, 1.19, ], 0.031), ([3, 2, 1, 0], [0.81, 1.2, 0.8], -0.05),
         ([1, 3, 0, 2], [1.0, 1.07, 0.93], 0.0499), ([2, 0], [1.2, 1.0, 0.85], 0.0), ([3], [1, 1, 1], -0.02), ([], [1, 1, 1], 0.0)]


@pytest.mark.parametrize("order,ratio,shift", CASES)
def test_torch_composition_matches_oracle(order, ratio, shift):
    img = _images(3, 24, 20, 1)
    got = imgops.ColorJitter.apply_ops(img, order, ratio, shift).numpy()
    ref = oracle_imgops.color_jitter_oracle(img.numpy(), order, ratio, shift)
    np.testing.assert_allclose(got, ref, atol=2e-6, rtol=0)


def test_draw_consumes_rng_like_forward():
    jit = imgops.ColorJitter(0.2, 0.2, 0.2, 0.05)
    torch.manual_seed(5)
    order, ratio, shift = jit.draw()
    assert sorted(order) == [0, 1, 2, 3] and all(0.8 <= r <= 1.2 for r in ratio) and abs(shift) <= 0.05
    torch.manual_seed(5)
    img = _images(2, 8, 8, 2)
    a = jit(img)
    assert torch.equal(a, imgops.ColorJitter.apply_ops(img, order, ratio, shift))


@pytest.mark.gpu
@pytest.mark.parametrize("order,ratio,shift", CASES + [(list(p), [0.9, 1.15, 1.1], 0.04) for p in itertools.permutations(range(4))][::5])
def test_fused_jitter_normalize_gpu(order, ratio, shift):
    img = _images(5, 64, 48, 3)

    class Fixed(imgops.ColorJitter):
        def draw(self):
            return order, ratio, shift

    norm = imgops.Normalize(MEAN, STD).cuda()
    out = imgops.jitter_normalize(img.cuda(), Fixed(0.2, 0.2, 0.2, 0.05), norm)
    assert out.is_contiguous(memory_format=torch.channels_last) and out.shape == img.shape
    ref = oracle_imgops.normalize_oracle(oracle_imgops.color_jitter_oracle(img.numpy(), order, ratio, shift), MEAN, STD)
    # tolerance: fp32 rounding through /std (x4.5) and the hue sector arithmetic (x6)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=0)


@pytest.mark.gpu
def test_fused_jitter_full_size_is_deterministic_and_bounded():
    img = torch.rand(32, 3, 256, 256, generator=torch.Generator().manual_seed(0)).cuda()
    jit, norm = imgops.ColorJitter(0.2, 0.2, 0.2, 0.05), imgops.Normalize(MEAN, STD).cuda()
    torch.manual_seed(11)
    a = imgops.jitter_normalize(img, jit, norm)
    torch.manual_seed(11)
    b = imgops.jitter_normalize(img, jit, norm)
    assert torch.equal(a, b)
    lo = (0 - torch.tensor(MEAN)) / torch.tensor(STD)
    hi = (1 - torch.tensor(MEAN)) / torch.tensor(STD)
    assert (a.amin((0, 2, 3)).cpu() >= lo - 1e-6).all() and (a.amax((0, 2, 3)).cpu() <= hi + 1e-6).all()
    # against the torch composition with the same draw
    torch.manual_seed(11)
    order, ratio, shift = jit.draw()
    ref = norm(imgops.ColorJitter.apply_ops(img, order, ratio, shift))
    assert (a - ref).abs().max().item() < 2e-5
