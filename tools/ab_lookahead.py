"""tools/ab_lookahead.py -- A/B inside ONE process: step time of the bench workload (B=32, 256x256) with and without the training
loop's one batch of look-ahead for the frozen DINO ViT (Trainer.step(data, next_data): the next batch's ViT pass is enqueued on the
side stream before this step's backward).  One ViT pass per step either way."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402
from scp_amd import synthetic as synth  # noqa: E402

tr, opts = bench.build_trainer("cuda:0", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda:0")
for _ in range(6):
    tr.step(data)


def timed(next_data, n=20):
    for _ in range(3):
        tr.step(data, next_data=next_data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(data, next_data=next_data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    print("no look-ahead   %.2f ms/step" % timed(None), flush=True)
    print("ViT look-ahead  %.2f ms/step" % timed(data), flush=True)
