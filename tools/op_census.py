"""tools/op_census.py -- attribute every GPU kernel of one training step to the stage of the step that
issued it (forward) or whose autograd node issued it (backward, linked through sequence numbers), and
split each stage into library/hand-written kernels vs small framework kernels.  Answers "where do the
~2500 small launches come from"."""
import collections
import functools
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile, record_function

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402
from scp_amd import correspondence, encoder, losses, mesh, pretrained_corr, renderer  # noqa: E402


def staged(owner, name, label):
    fn = getattr(owner, name)

    @functools.wraps(fn)
    def wrapper(*a, **k):
        with record_function("stage:" + label):
            return fn(*a, **k)
    setattr(owner, name, wrapper)


staged(encoder.Encoder, "forward", "encoder")
staged(renderer.Renderer, "render_all", "render")
staged(correspondence.Correspondence, "match", "fv-match")
staged(correspondence.Correspondence, "compute_rotation_cycle_loss", "rotation-cycle")
staged(pretrained_corr.PretrainedCorrespondence, "compute_cycle_loss", "dino-cycle")
staged(pretrained_corr.PretrainedCorrespondence, "prefetch_features", "dino-vit")
for fn in [n for n in dir(losses) if n.startswith("compute_")]:
    staged(losses, fn, "losses")
staged(mesh.CanonicalMesh, "compute_symmetry_loss", "symmetry")
staged(mesh.CanonicalMesh, "get_texture", "texture-sample")

tr, opts = bench.build_trainer("cuda", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
for _ in range(4):
    tr.step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(data)
    torch.cuda.synchronize()
evts = prof.events()


def ancestors(e):
    while e is not None:
        yield e
        e = e.cpu_parent


seq_stage = {}
for e in evts:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and not e.name.startswith("autograd::"):
        for a in ancestors(e):
            if a.name.startswith("stage:"):
                seq_stage.setdefault(e.sequence_nr, a.name[6:])
                break

BIG = ("conv", "gemm", "Cijk", "igemm", "raster", "vit_attention", "add_layernorm", "cols_", "rows_", "dual_backward",
       "nearest", "BatchNorm", "batch_norm", "face_setup", "mm", "addmm", "bmm")
table = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
small_ops = collections.defaultdict(lambda: [0, 0.0])
for e in evts:
    dt = e.self_device_time_total
    if dt <= 0 or e.device_type != torch.autograd.DeviceType.CPU:
        continue
    stage, phase = None, "fwd"
    for a in ancestors(e):
        if a.name.startswith("stage:"):
            stage = a.name[6:]
            break
        if a.name.startswith("autograd::engine::evaluate_function"):
            phase = "bwd"
            stage = seq_stage.get(a.sequence_nr, "?")
            break
        if a.name.startswith("Optimizer.step"):
            stage = "optimizer"
            break
    stage = (stage or "step-glue") + ":" + phase
    nk = len(e.kernels)
    big = any(b in e.name for b in BIG)
    row = table[stage]
    row[0 if big else 2] += nk
    row[1 if big else 3] += dt / 1e3
    if not big:
        key = (stage, e.name, str(e.input_shapes)[:90])
        small_ops[key][0] += nk
        small_ops[key][1] += dt / 1e3
print("%-26s %8s %9s | %8s %9s" % ("stage", "big n", "big ms", "small n", "small ms"))
tot = [0, 0.0, 0, 0.0]
for k, v in sorted(table.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    print("%-26s %8d %9.2f | %8d %9.2f" % (k, *v))
    tot = [a + b for a, b in zip(tot, v)]
print("%-26s %8d %9.2f | %8d %9.2f" % ("TOTAL", *tot))
print("\ntop small ops")
for (stage, name, shp), (n, ms) in sorted(small_ops.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%7.3f ms x%-3d %-22s %-32s %s" % (ms, n, stage, name[:32], shp))

# ---- framework (aten::) launches by call site: what the ~500 small launches of a step are ------------------------------------------------
aten = collections.defaultdict(lambda: [0, 0.0])
for (stage, name, shp), (n, ms) in small_ops.items():
    if name.startswith("aten::") or name.startswith("hipMemcpy") or name.startswith("Memcpy") or name.startswith("Memset"):
        aten[(stage, name)][0] += n
        aten[(stage, name)][1] += ms
tot_n, tot_ms = sum(v[0] for v in aten.values()), sum(v[1] for v in aten.values())
print("\naten:: launches of the step: %d, %.2f ms of device time; by (stage, op), sorted by launches" % (tot_n, tot_ms))
for (stage, name), (n, ms) in sorted(aten.items(), key=lambda kv: -kv[1][0])[:90]:
    print("x%-4d %7.3f ms  %-22s %s" % (n, ms, stage, name))
by_stage = collections.defaultdict(lambda: [0, 0.0])
for (stage, name), (n, ms) in aten.items():
    by_stage[stage][0] += n
    by_stage[stage][1] += ms
print("\naten:: launches by stage")
for stage, (n, ms) in sorted(by_stage.items(), key=lambda kv: -kv[1][0]):
    print("x%-4d %7.3f ms  %s" % (n, ms, stage))
