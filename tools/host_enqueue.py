"""tools/host_enqueue.py -- is the step host-bound?  Times the CPU side of each phase of Trainer.step (forward enqueue, backward
enqueue, gradient collection + optimizer) without synchronising inside the step, next to the synchronised step time of the same
loop.  If the phases' host time adds up to the step time, the GPU is waiting for launches and fewer / cheaper launches is the lever."""
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
torch.cuda.synchronize()

N = 20
acc = {"prepare": 0.0, "forward": 0.0, "backward": 0.0, "collect+optim": 0.0}
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter()
    tr.model.iters = tr.iteration
    tr.grads.prepare()
    t1 = time.perf_counter()
    total, aux = tr.model(data)
    t2 = time.perf_counter()
    total.mean().backward()
    t3 = time.perf_counter()
    tr.collect_grad()
    tr.optim.step(tr.iteration)
    tr.iteration += 1
    t4 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[k] += v
host = time.perf_counter() - t_all
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print("host enqueue per step: " + ", ".join("%s %.2f ms" % (k, v / N * 1e3) for k, v in acc.items()))
print("host loop %.2f ms/step, synchronised %.2f ms/step (queue drained %.2f ms after the last enqueue)"
      % (host / N * 1e3, wall / N * 1e3, (wall - host) * 1e3))

# pure host cost: each step starts on an EMPTY queue (synchronised before), so no enqueue can block on a full queue
costs = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(data)
    costs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
costs.sort()
print("host time of one step enqueued on an empty queue: median %.2f ms, min %.2f ms" % (costs[len(costs) // 2] * 1e3, costs[0] * 1e3))
