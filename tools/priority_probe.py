"""tools/priority_probe.py -- does HIP stream priority on the side streams (frozen ViT / rotation-cycle branch) move the step time?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402


def timed(tr, data, n=30, w=8):
    for _ in range(w):
        tr.step(data)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


tr, opts = bench.build_trainer("cuda", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
print("default priorities: %.2f ms" % timed(tr, data))
for name, (pd, pc) in (("ViT high", (-1, 0)), ("cycle high", (0, -1)), ("both high", (-1, -1))):
    tr.model.pretrain_corr_net._side_stream = torch.cuda.Stream(priority=pd)
    tr.model._cycle_stream = torch.cuda.Stream(priority=pc)
    print("%-12s %.2f ms" % (name, timed(tr, data)))
