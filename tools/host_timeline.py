"""tools/host_timeline.py -- where does the HOST spend a training step, and does it ever block on the device?
Runs 30 un-synchronised steps and prints the mean host time of each stage of Trainer.step (perf_counter, no device sync)
next to the device time per step.  If a stage's host time is close to the device step time, the host is blocking there."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from scp_amd import synthetic  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
data = synthetic.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
for _ in range(6):
    tr.step(data)
torch.cuda.synchronize()
marks = {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter()
        out = fn(*a, **k)
        marks.setdefault(label, []).append(time.perf_counter() - t)
        return out
    setattr(obj, name, w)


m = tr.model
wrap(tr.grads, "prepare", "grads.prepare")
wrap(m.encoder, "forward", "  encoder fwd (enqueue)")
wrap(m.corr_net, "match", "  corr match")
wrap(m.renderer, "render_all", "  render_all")
wrap(m.mesh, "compute_symmetry_loss", "  symmetry loss")
wrap(m.pretrain_corr_net, "compute_cycle_loss", "  DINO cycle loss")
wrap(m.corr_net, "compute_rotation_cycle_loss", "  rotation cycle")
wrap(m, "forward", "model forward")
wrap(tr, "collect_grad", "collect_grad")
wrap(tr.optim, "step", "optim.step")
_bw = torch.Tensor.backward


def bw(self, *a, **k):
    t = time.perf_counter()
    out = _bw(self, *a, **k)
    marks.setdefault("backward", []).append(time.perf_counter() - t)
    return out


torch.Tensor.backward = bw
N = 30
t0 = time.perf_counter()
stamps = []
for _ in range(N):
    tr.step(data)
    stamps.append(time.perf_counter())
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host loop %.2f ms/step, device-drained %.2f ms/step" % (1e3 * t_host / N, 1e3 * t_all / N))
for k, v in marks.items():
    print("%-28s %7.2f ms  (max %.2f)" % (k, 1e3 * sum(v) / len(v), 1e3 * max(v)))
d = [1e3 * (b - a) for a, b in zip(stamps, stamps[1:])]
print("step-to-step host intervals: first 10:", ["%.1f" % x for x in d[:10]], " last 5:", ["%.1f" % x for x in d[-5:]])
