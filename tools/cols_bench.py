"""tools/cols_bench.py -- column soft-argmax over the pooled scores (a9: [32, 1024, 642], no masks), forward and backward kernels of
csrc/corr.hip, HIP events around 20 calls each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import corr_ops  # noqa: E402

N, P, Q = 32, 1024, 642
g = torch.Generator().manual_seed(0)
s = torch.randn(N, P, Q, generator=g).cuda()
grid = (torch.rand(2, P, generator=g) * 2 - 1).cuda()
w = torch.randn(N, 2, Q, generator=g).cuda()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out, stats = corr_ops.cols_forward(s, None, None, grid, 10.)
t_f = timed(lambda: corr_ops.cols_forward(s, None, None, grid, 10.))
t_b = timed(lambda: corr_ops.dual_backward(s, None, None, None, (stats, out, w, grid, 10.), None))
print("cols_softargmax [%d,%d,%d]: forward %.3f ms, backward %.3f ms (84 MB read forward; 84 MB read + 84 MB written backward)" % (N, P, Q, t_f, t_b))
