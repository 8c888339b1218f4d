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


@pytest.mark.parametrize("angle", [17.0, 45.0, 133.7, 271.3, 359.0])
def test_rotate_at_arbitrary_angles_follows_the_published_torchvision_algorithm(angle):
    """VERDICT r5 missing #4: the rotation-cycle augmentation at angles that are not multiples of 90 degrees (correspondence.py:82-89 draws
    U(0, 360)).  torchvision is absent, so the comparator is oracle/imgops.rotate_oracle -- torchvision 0.11's published tensor algorithm
    (inverse affine matrix, pixel-centre base grid, grid_sample with zero padding) restated in numpy: parity UNPINNED to an executed
    torchvision, pinned to its formulas.  Bilinear: 2e-5 everywhere.  Nearest (masks, the ground-truth grid): exact wherever the source
    coordinate is not within 1e-3 of a rounding tie (two float32 evaluation orders of the same affine map may round a tie differently)."""
    g = torch.Generator().manual_seed(int(angle * 10))
    img = torch.rand(2, 3, 64, 64, generator=g)
    mask = (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float()
    got_b = imgops.rotate(img, angle, "bilinear").numpy()
    ref_b, _ = oracle_imgops.rotate_oracle(img.numpy(), angle, "bilinear")
    assert np.abs(got_b - ref_b).max() <= 2e-5
    got_n = imgops.rotate(mask, angle, "nearest").numpy()
    ref_n, (ix, iy) = oracle_imgops.rotate_oracle(mask.numpy(), angle, "nearest")
    tie = (np.abs(ix - np.floor(ix) - 0.5) < 1e-3) | (np.abs(iy - np.floor(iy) - 0.5) < 1e-3)
    assert np.array_equal(got_n[:, :, ~tie], ref_n[:, :, ~tie])
    assert tie.mean() < 0.05 and 0.2 < got_n.mean() < 0.6, "the comparison must cover nearly all pixels of a half-filled mask (45 degrees: the diagonals are ties)"


@pytest.mark.gpu
def test_rotate_at_arbitrary_angles_on_the_gpu_equals_the_cpu_path():
    """the same ATen grid path on the MI355X: bilinear to 2e-5 of the CPU path and of the oracle, nearest exact off the ties"""
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, 64, 64, generator=g)
    for angle in (17.0, 133.7, 271.3):
        cpu = imgops.rotate(img, angle, "bilinear")
        gpu = imgops.rotate(img.cuda(), angle, "bilinear").cpu()
        # 5e-6 observed: the two back ends round the sampling coordinate differently in the last bit, times the image gradient
        assert float((cpu - gpu).abs().max()) <= 2e-5
        ref_b, _ = oracle_imgops.rotate_oracle(img.numpy(), angle, "bilinear")
        assert np.abs(gpu.numpy() - ref_b).max() <= 2e-5
        ref_n, (ix, iy) = oracle_imgops.rotate_oracle(img.numpy(), angle, "nearest")
        tie = (np.abs(ix - np.floor(ix) - 0.5) < 1e-3) | (np.abs(iy - np.floor(iy) - 0.5) < 1e-3)
        gpu_n = imgops.rotate(img.cuda(), angle, "nearest").cpu().numpy()
        assert np.array_equal(gpu_n[:, :, ~tie], ref_n[:, :, ~tie])
