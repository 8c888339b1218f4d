"""tools/phase_timeline.py -- GPU timeline of the step's phases WITHOUT a profiler: HIP events recorded on the streams the phases
run on (forward: around the wrapped calls; backward: from gradient hooks on the phases' inputs / outputs), read back after the run.
rocprofv3's per-launch interception makes this workload host-bound (50 ms steps), so its timeline is not the free-running one; a few
dozen events per step do not disturb anything.  Prints, per mark, the median time since the step's first event over the measured steps.  --lookahead: with the training loop's
one batch of look-ahead for the frozen ViT (bench.py's default)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402
from scp_amd import fused_losses, synthetic as synth  # noqa: E402

tr, opts = bench.build_trainer("cuda:0", 1)
model = tr.model
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda:0")

MARKS = None   # list of (label, event) for the current step
LOOKAHEAD = "--lookahead" in sys.argv


def mark(label):
    if MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        MARKS.append((label, ev))


def hook_grad(t, label):
    if MARKS is not None and isinstance(t, torch.Tensor) and t.requires_grad:
        t.register_hook(lambda g: mark(label))


def wrap(obj, name, label, outs=lambda r: r if isinstance(r, tuple) else (r,)):
    fn = getattr(obj, name)

    def inner(*a, **k):
        mark(label + ":fwd>")
        for i, t in enumerate(a):
            hook_grad(t, "%s:bwd-done(in%d)" % (label, i))
        r = fn(*a, **k)
        mark(label + ":fwd<")
        for i, t in enumerate(outs(r)):
            hook_grad(t, "%s:bwd-start(out%d)" % (label, i))
        return r
    setattr(obj, name, inner)


wrap(model.encoder, "encode_img", "encode_img")
wrap(model.encoder, "forward", "encoder")
wrap(model.corr_net, "match", "match")
wrap(model.corr_net, "compute_rotation_cycle_loss", "rot_cycle")
wrap(model.renderer, "render_depth_group", "render_depth")
wrap(model.renderer, "render_texture_raw", "render_tex")
wrap(model.pretrain_corr_net, "compute_cycle_loss", "dino_cycle")
wrap(model.pretrain_corr_net, "prefetch_features", "dino_prefetch")
wrap(model.pretrain_corr_net.net, "forward", "vit")
wrap(fused_losses, "depth_group_losses", "losses")
wrap(model.mesh, "get_texture", "get_texture")
wrap(model.mesh, "compute_symmetry_loss", "symmetry")

orig_step = tr.step
for _ in range(6):
    tr.step(data)
torch.cuda.synchronize()

records = []
for _ in range(12):
    MARKS = []
    mark("step>")
    total, aux = None, None
    tr.model.iters = tr.iteration
    tr.grads.prepare()
    total, aux = tr.model(data)
    mark("forward-enqueued")
    if LOOKAHEAD:      # Trainer.step(data, next_data=data): the next batch's DINO pass goes to the side stream before the backward
        tr.model.pretrain_corr_net.prefetch_features(data[0], data[1])
    total.mean().backward()
    mark("backward<")
    tr.collect_grad()
    tr.optim.step(tr.iteration)
    tr.iteration += 1
    mark("step<")
    records.append(MARKS)
MARKS = None
torch.cuda.synchronize()

# labels can repeat inside a step (encode_img twice, hooks on several inputs): number them in order of recording
rows = {}
for marks in records[2:]:
    t0 = marks[0][1]
    seen = {}
    for label, ev in marks:
        n = seen.get(label, 0)
        seen[label] = n + 1
        rows.setdefault((label, n), []).append(t0.elapsed_time(ev))
out = []
for (label, n), ts in rows.items():
    ts.sort()
    out.append((ts[len(ts) // 2], label + ("#%d" % n if n else "")))
out.sort()
for t, label in out:
    print("%8.2f ms  %s" % (t, label))
