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
# ---- coarse per-stage GPU time (HIP events around the stage entry points) -------------------------
stages = {}


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        stages.setdefault(label, []).append((e0, e1))
        return out
    setattr(obj, name, wrapper)


m = tr.model
timed(m.encoder, "forward", "fwd encoder (1st pass)")
timed(m.corr_net, "match", "fwd feature<->vertex match")
timed(m.renderer, "render_all", "fwd 4 render passes")
timed(m.mesh, "compute_symmetry_loss", "fwd symmetry loss")
timed(m.pretrain_corr_net, "compute_cycle_loss", "fwd DINO cycle loss (incl. ViT)")
timed(m.pretrain_corr_net.net, "forward", "  of which DINO ViT")
timed(m.corr_net, "compute_rotation_cycle_loss", "fwd rotation cycle (2nd encoder pass)")
timed(m, "forward", "FORWARD total")
timed(tr, "collect_grad", "clip + nan guard")
timed(tr.optim, "step", "AdamW + scheduler")
e_all0, e_all1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e_all0.record()
for _ in range(5):
    tr.step(data)
e_all1.record()
torch.cuda.synchronize()
tot = e_all0.elapsed_time(e_all1) / 5
print("GPU-timeline step: %.2f ms" % tot)
acc = 0
for label, evs in stages.items():
    ms = sum(a.elapsed_time(b) for a, b in evs) / 5
    print("  %-42s %7.2f ms" % (label, ms))
fwd = sum(a.elapsed_time(b) for a, b in stages["FORWARD total"]) / 5
rest = sum(sum(a.elapsed_time(b) for a, b in stages[k]) / 5 for k in ("clip + nan guard", "AdamW + scheduler"))
print("  %-42s %7.2f ms" % ("BACKWARD (+zero_grad) = total - fwd - rest", tot - fwd - rest))
sys.stdout.flush()
if "--ops" not in sys.argv:
    sys.exit(0)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -r.device_time_total)[:130]
for r in rows:
    print("%9.3f ms  x%-4d %-60s %s" % (r.device_time_total / 1e3, r.count, r.key[:60], str(r.input_shapes)[:110]))
print("---- per-op totals (self device time) ----")
rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)[:60]
for r in rows:
    print("%9.3f ms  x%-5d %s" % (r.self_device_time_total / 1e3, r.count, r.key[:90]))
