"""tools/step_ablate.py -- what is each branch of the step worth in wall-clock on the GPU?  Times the
training step with individual branches short-circuited (results are then wrong -- timing only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    sys.path.insert(0, p)
import bench  # noqa: E402
from scp_amd import synthetic as synth  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")


def timeit(n=10):
    for _ in range(4):
        tr.step(data)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("full step                        %.2f ms" % timeit())
net = tr.model.pretrain_corr_net.net
feats = net(data[0])
orig_fwd = net.forward
net.forward = lambda img: feats
print("DINO ViT replaced by a constant   %.2f ms" % timeit())
net.forward = orig_fwd
m = tr.model
m.overlap_dino = False
print("DINO on the main stream           %.2f ms" % timeit())
m.overlap_dino = True
m.overlap_rotation_cycle = False
print("rotation cycle on the main stream %.2f ms" % timeit())
m.overlap_rotation_cycle = True
orig_cycle = m.corr_net.compute_rotation_cycle_loss
m.corr_net.compute_rotation_cycle_loss = lambda *a, **k: (a[2].sum() * 0,)
print("rotation cycle removed            %.2f ms" % timeit())
m.corr_net.compute_rotation_cycle_loss = orig_cycle
orig_render = m.renderer.render_all
cache = {}


def cached_render(*a, **k):
    if "out" not in cache:
        cache["out"] = [t.detach() if torch.is_tensor(t) else t for t in orig_render(*a, **k)]
    return cache["out"]


m.renderer.render_all = cached_render
print("render passes replaced by consts  %.2f ms" % timeit())
m.renderer.render_all = orig_render
# encoder convolutions: how much of the step is the ResNet / decoder (both passes)?  Replace the first encoder pass' backbone+featnet
# output by a constant (keeps the pose / shape heads)
enc = m.encoder
orig_enc = enc.forward
enc_cache = {}


def cached_enc(*a, **k):
    if "out" not in enc_cache:
        enc_cache["out"] = [t.detach() if torch.is_tensor(t) else t for t in orig_enc(*a, **k)]
    return enc_cache["out"]


enc.forward = cached_enc
print("1st encoder pass replaced by const %.2f ms (no encoder backward either)" % timeit())
m.corr_net.compute_rotation_cycle_loss = lambda *a, **k: (a[2].sum() * 0,)
print("... and rotation cycle removed     %.2f ms (= no convolution at all)" % timeit())
net.forward = lambda img: feats
print("... and DINO constant              %.2f ms" % timeit())
