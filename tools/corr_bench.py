"""tools/corr_bench.py -- Correspondence.match + the 2x2 pooling, forward + backward at the bench size (B=32, 64x64 map, 642 or
995 vertices, 64 channels): fused kernels (csrc/corr_fused.hip, scores never stored) vs the round-1 path (rocBLAS scores +
reduction kernels of csrc/corr.hip)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import ops  # noqa: E402
from scp_amd.correspondence import Correspondence  # noqa: E402
from scp_amd.flags import Options  # noqa: E402

opts = Options("laptop_wild6d", train=True)
B = 32
for V in (642, 995):
    g = torch.Generator().manual_seed(V)
    img = torch.nn.functional.normalize(torch.randn(B, 64, 4096, generator=g), 2, 1).cuda().requires_grad_(True)
    mesh = torch.nn.functional.normalize(torch.randn(B, V, 64, generator=g), 2, 2).cuda().requires_grad_(True)
    pred_v = torch.randn(B, V, 3, generator=g).cuda()
    mask = (torch.rand(B, 256, 256, generator=g) > 0.3).float().cuda()
    for fuse in (True, False):
        corr = Correspondence(opts, "cuda")
        corr.fuse_scores = fuse

        def step():
            pc, match, imatch, _ = corr.match(img, mesh, mask, pred_v)
            pooled = ops.pool2x2_scores(pc, 64, 64)
            ((pooled * 1e-3).sum() + match.sum() + imatch.sum()).backward()
            img.grad = mesh.grad = None

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step()
        e1.record()
        torch.cuda.synchronize()
        print("V=%d %s: match + pool, forward + backward %.3f ms" % (V, "fused  " if fuse else "unfused", e0.elapsed_time(e1) / 20))
