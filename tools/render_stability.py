"""tools/render_stability.py -- is Renderer.render_all on the G2 fixture bit-stable within a process and across process states?
(tests/test_render_golden.py passed alone and failed inside the full suite; this tells an in-process race from state-dependent inputs)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")]
import golden_io
import scp_amd.dino as dino
from scp_amd.flags import Options
from scp_amd.model import MeshNet
from scp_amd import losses

dino.ALLOW_RANDOM_INIT = True
d = golden_io.load("render_all_bottle_b2")
step = golden_io.load("step_laptopflags_bottle_b2x2")
opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True)
model = MeshNet(opts, prior=(step["prior_verts"], step["prior_faces"])).cuda()
t = lambda k: torch.tensor(d[k], device="cuda")
faces = model.mesh.faces[None].expand(2, -1, -1)
hx = lambda x: int(x.contiguous().view(torch.int32).long().sum())


def run():
    with torch.no_grad():
        proj = losses.project_for_render(t("pred_v"), t("foc_crop"), t("pp_crop"), t("rotation"), t("translation"))
        outs = model.renderer.render_all(t("pred_v"), faces, t("tex"), t("foc_crop"), t("pp_crop"), t("rotation"), t("translation"), None)
    return (hx(proj),) + tuple(hx(o) for o in outs), outs


def report(tag, n=100):
    ref, outs = run()
    bad = sum(run()[0] != ref for _ in range(n))
    dr = outs[2].double().cpu()
    print("%-34s hashes %s...  unstable passes %d/%d  depth_render |sum| %.6f (fixture %.6f)  mask sum %.6f" % (
        tag, str(ref[:3]), bad, n, float(dr.abs().sum()), float(d["stats_depth_render"][1]), float(outs[0].double().sum())), flush=True)
    return ref


a = report("fresh process")
from scp_amd.trainer import Trainer
from scp_amd import synthetic
tr = Trainer(Options("laptop_wild6d", batch_size=1, repeat=2, train=True, total_iters=10), prior=synthetic.bottle_like(2), device="cuda")
b = report("after constructing a Trainer")
tr.step(synthetic.make_batch(1, 2, 256, seed=0, device="cuda"))
torch.cuda.synchronize()
c = report("after one training step")
junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(64)]
del junk
e = report("after filling freed memory with NaN")
print("same across states:", a == b == c == e)
