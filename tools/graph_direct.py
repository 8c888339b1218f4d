"""one graph-mode training run in THIS process (see tools/graph_variants.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import torch
import scp_amd.dino as dino
from scp_amd.flags import Options
from scp_amd.trainer import Trainer
from scp_amd import synthetic
dino.ALLOW_RANDOM_INIT = True
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opts = Options("laptop_wild6d", batch_size=bs, repeat=4, train=True, total_iters=100)
torch.manual_seed(0)
tr = Trainer(opts, prior=synthetic.bottle_like(3), device="cuda", graphs=True)
batches = [synthetic.make_batch(bs, 4, 256, seed=30 + i, device="cuda") for i in range(nsteps)]
for i, d in enumerate(batches):
    total, aux, _ = tr.step(d)
    print("step", i, float(total), flush=True)
torch.cuda.synchronize()
print("OK")
