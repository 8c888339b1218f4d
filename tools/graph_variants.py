"""tools/graph_variants.py -- which configuration breaks the HIP-graph capture of the training step's segments?  Each variant in its
own process (a crash inside hipStreamEndCapture takes the interpreter down)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "self-corr-pose_amd")); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
import scp_amd.dino as dino
from scp_amd.flags import Options
from scp_amd.trainer import Trainer
from scp_amd import synthetic
dino.ALLOW_RANDOM_INIT = True
bs, rep, ident, angle, look, same, how = %s
if how == "bench":
    import bench
    os.environ["SCP_GRAPHS"] = "1"
    tr, opts = bench.build_trainer("cuda:0", 1, bs, rep)
else:
    kw = dict(total_iters=100) if how != "long" else {}
    if how == "novis":
        kw["vis_freq"] = 10 ** 9
    opts = Options("laptop_wild6d", batch_size=bs, repeat=rep, train=True, **kw)
    torch.manual_seed(0)
    tr = Trainer(opts, prior=synthetic.bottle_like(3), device="cuda:0" if how == "cuda0" else "cuda", graphs=True)
if ident:
    tr.model.encoder.random_jitter = torch.nn.Identity()
if angle is not None:
    tr.model.rotation_angle = angle
batches = [synthetic.make_batch(bs, rep, 256, seed=30 + i, device="cuda") for i in range(5)]
if same:
    batches = [batches[0]] * 5
for i, d in enumerate(batches):
    nxt = batches[i + 1] if (look and i + 1 < len(batches)) else None
    total, aux, _ = tr.step(d, next_data=nxt)
torch.cuda.synchronize()
print("OK loss %%.6f" %% float(total))
'''
VARIANTS = {"B32 bench.build_trainer": (8, 4, False, None, False, True, "bench"), "B32 device cuda:0": (8, 4, False, None, False, True, "cuda0"),
            "B32 vis_freq off": (8, 4, False, None, False, True, "novis"), "B32 default total_iters": (8, 4, False, None, False, True, "long"),
            "B32 as before": (8, 4, False, None, False, True, "")}
for name, cfg in VARIANTS.items():
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT, ROOT, repr(cfg))], capture_output=True, text=True)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    print("%-32s rc %4d  %s" % (name, r.returncode, tail if r.returncode == 0 else (r.stderr.strip().splitlines() or ["?"])[-1][:160]), flush=True)
