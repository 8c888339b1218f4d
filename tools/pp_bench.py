"""tools/pp_bench.py -- pixel <-> pixel soft-argmax of the rotation-cycle loss (a10) at the bench size (N = 32, 64 channels, 32 x 32
half-resolution maps): fused kernels (csrc/corr_pp.hip, scores never stored) vs the two-step path (library GEMMs + csrc/corr.hip),
forward + backward."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import ops  # noqa: E402

N, C, P = 32, 64, 1024
g = torch.Generator().manual_seed(1)
src = torch.nn.functional.normalize(torch.randn(N, C, P, generator=g), 2, 1).cuda().requires_grad_(True)
tgt = torch.nn.functional.normalize(torch.randn(N, C, P, generator=g), 2, 1).cuda().requires_grad_(True)
yy, xx = torch.meshgrid(torch.arange(32), torch.arange(32), indexing="ij")
mask = (((yy - 16) ** 2 + (xx - 16) ** 2) < 12 ** 2).float().reshape(1, P).repeat(N, 1).cuda()
grid = (torch.rand(N, 2, P, generator=g) * 2 - 1).cuda()
w = torch.randn(N, 2, P, generator=g).cuda()


def step(fused):
    if fused:
        out = ops.pixel_pixel_softargmax(src, tgt, mask, mask, grid, 10.)
    else:
        out = ops.cols_softargmax(src.transpose(1, 2).bmm(tgt), mask, mask, grid, 10.)
    (out * w).sum().backward()
    src.grad = tgt.grad = None


for fused in (True, False):
    for _ in range(5):
        step(fused)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        step(fused)
    e1.record()
    torch.cuda.synchronize()
    print("pixel_pixel_softargmax %s: forward + backward %.3f ms" % ("fused   " if fused else "two-step", e0.elapsed_time(e1) / 20))
