"""tools/step_profile.py -- torch.profiler view of one full training step at the bench size:
per-op GPU time (grouped by input shape) and a coarse per-stage wall-clock breakdown."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
for _ in range(4):
    tr.step(data)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    tr.step(data)
torch.cuda.synchronize()
print("step: %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -r.device_time_total)[:45]
for r in rows:
    print("%9.3f ms  x%-4d %-60s %s" % (r.device_time_total / 1e3, r.count, r.key[:60], str(r.input_shapes)[:110]))
