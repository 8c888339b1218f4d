"""tools/batch_scaling.py -- images/s of the full training step on ONE MI355X as the per-GPU batch grows (the reference's
8 videos x 4 frames = 32 is the headline; 288 GB of HBM allow far more)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402

for bs in (8, 16, 32, 64):
    tr, opts = bench.build_trainer("cuda", 1, batch_size=bs, repeat=4)
    data = synth.make_batch(bs, 4, opts.img_size, seed=100, device="cuda")
    for _ in range(6):
        tr.step(data)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 12
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    print("B = %3d images (batch_size %d x repeat 4): %.1f ms/step, %.0f images/s, peak memory %.1f GB" % (
        bs * 4, bs, ms, bs * 4 / ms * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
    del tr, data
    torch.cuda.empty_cache()
