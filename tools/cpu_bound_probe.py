"""tools/cpu_bound_probe.py -- is the step host-bound?  Time for Trainer.step() to RETURN (host enqueue
time) right after a device sync, vs the device time of the same step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
for _ in range(5):
    tr.step(data)
host, dev = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    dev.append((t2 - t0) * 1e3)
print("host enqueue per step %.1f ms (min %.1f) | step incl. device drain %.1f ms" % (
    sum(host) / len(host), min(host), sum(dev) / len(dev)))
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(3):
    tr.step(data)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
