"""tools/bn_bench.py -- fused BatchNorm(+skip)+ReLU HIP op vs the MIOpen/ATen composition at the encoder's
activation shapes (B=32, 256x256 input): fwd+bwd time and the per-kernel table of the fused path."""
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    sys.path.insert(0, p)
from scp_amd import fused_bn  # noqa: E402

SHAPES = [(32, 64, 128, 128), (32, 64, 64, 64), (32, 128, 32, 32), (32, 256, 16, 16), (32, 512, 8, 8)]
dev = torch.device("cuda")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for shape in SHAPES:
    for skip in (False, True):
        x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        s = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) if skip else None
        dy = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
        bn = nn.BatchNorm2d(shape[1]).to(dev)
        for p in bn.parameters():
            p.requires_grad_(False)

        def fused():
            fused_bn.bn_act(x, bn, s, relu=True).backward(dy)
            x.grad = None

        def stock():
            fused_bn._composition(x, bn, s, True).backward(dy)
            x.grad = None

        mb = x.numel() * 4 / 1e6
        tf, ts = timeit(fused), timeit(stock)
        passes = 11 if skip else 8
        print("%-20s skip=%d  fused %.3f ms (%.0f GB/s at %d passes)  stock %.3f ms" % (
            shape, skip, tf, passes * mb / tf, passes, ts), flush=True)

from torch.profiler import ProfilerActivity, profile  # noqa: E402
x = torch.randn(SHAPES[0], device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
dy = torch.randn_like(x)
bn = nn.BatchNorm2d(64).to(dev)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        fused_bn.bn_act(x, bn, None, relu=True).backward(dy)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10, max_name_column_width=70))
