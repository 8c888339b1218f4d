"""tools/soak.py -- 300 training steps on rotating synthetic batches: device memory must stay flat, losses finite, step time
stable (leak / drift check)."""
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
batches = [synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=200 + i, device="cuda") for i in range(4)]
marks = {}
t0 = time.perf_counter()
for i in range(300):
    total, aux, grad = tr.step(batches[i % 4])
    if i in (19, 99, 199, 299):
        torch.cuda.synchronize()
        marks[i] = (torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20, float(total.detach()),
                    (time.perf_counter() - t0))
        t0 = time.perf_counter()
for i, (alloc, res, loss, dt) in marks.items():
    print("step %3d: allocated %.0f MiB reserved %.0f MiB loss %.5f  (%.1f s since last mark)" % (i + 1, alloc, res, loss, dt))
a = [m[0] for m in marks.values()]
assert max(a[1:]) - min(a[1:]) < 64, "device memory drifts"
print("ok")
