"""tests/synth.py -- the synthetic training batch of SURVEY.md 8(d) (own code, seeded).

img ~ U(0,1); mask = filled ellipse around the image centre; depth = mask * (500 + 40*smooth noise);
crop intrinsics around f = 755 px (256-px crop), principal point near the centre.  Returned as the
12-tuple MeshNet.forward consumes (after Trainer.batch_reshape's NDC conversion)."""
import numpy as np
import torch
import torch.nn.functional as F


def make_batch(batch_size, repeat, img_size=256, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    n = batch_size * repeat
    s = img_size
    img = torch.rand(n, 3, s, s, generator=g)
    yy, xx = torch.meshgrid(torch.arange(s, dtype=torch.float32), torch.arange(s, dtype=torch.float32), indexing="ij")
    ax = (85 + 22 * torch.rand(n, generator=g)) * s / 256
    ay = (85 + 22 * torch.rand(n, generator=g)) * s / 256
    c = (s - 1) / 2
    mask = (((xx[None] - c) / ax[:, None, None]) ** 2 + ((yy[None] - c) / ay[:, None, None]) ** 2 <= 1).float()
    noise = F.avg_pool2d(torch.randn(n, 1, s, s, generator=g), 15, 1, 7)[:, 0]
    depth = mask * (500 + 40 * noise * 7)
    foc_px = 755 * s / 256 * (0.9 + 0.2 * torch.rand(n, 2, generator=g))
    pp_px = s / 2 + (50 * torch.rand(n, 2, generator=g) - 25) * s / 256
    pp_crop = pp_px / (s / 2.) - 1.
    foc_crop = foc_px / (s / 2.)
    z2 = torch.zeros(n, 2)
    data = (img, mask, depth, None, z2, z2, foc_px, foc_crop, pp_px, pp_crop, torch.arange(n), None)
    return tuple(t.to(device) if torch.is_tensor(t) else t for t in data)
