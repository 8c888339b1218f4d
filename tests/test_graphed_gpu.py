"""scp_amd/graphed.py (opt-in, SCP_GRAPHS=1): HIP-graph replay of the two encoder passes (forward + backward) and of the frozen ViT
inside the full training step, at the workload it is validated on -- BASELINE's B = 32 per GPU.  The graph trainer runs in a CHILD
PROCESS: a failing hipStreamEndCapture takes the interpreter down (observed at B <= 8 on this ROCm stack, tools/graph_variants.py --
which is why the feature is opt-in), and that must fail this test, not the test session."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
ROOT = %r
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import scp_amd.dino as dino
from scp_amd.flags import Options
from scp_amd.trainer import Trainer
from scp_amd import synthetic
dino.ALLOW_RANDOM_INIT = True
graphs = %r
opts = Options("laptop_wild6d", batch_size=8, repeat=4, train=True, total_iters=100)
torch.manual_seed(0)
tr = Trainer(opts, prior=synthetic.bottle_like(3), device="cuda", graphs=graphs)
tr.model.rotation_angle = 90.0
tr.model.encoder.random_jitter = torch.nn.Identity()
torch.manual_seed(1)
batches = [synthetic.make_batch(8, 4, 256, seed=30 + i, device="cuda") for i in range(5)]
hist = []
for i, d in enumerate(batches):
    total, aux, _ = tr.step(d, next_data=batches[i + 1] if i + 1 < len(batches) else None)
    hist.append({k: float(v.detach()) for k, v in aux.items()})
captured = None
if graphs:
    segs = tr.model.encoder._graph_segments
    captured = bool(segs["full"].graphs is not None and segs["half"].graphs is not None and tr.model.pretrain_corr_net._vit_graph.graph is not None)
print("RESULT " + json.dumps({"hist": hist, "captured": captured}))
'''


def _run(graphs):
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, graphs)], capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, "child (graphs=%s) exited with %s: %s" % (graphs, r.returncode, r.stderr.strip().splitlines()[-3:])
    return json.loads(line[-1][7:])


def test_trainer_steps_with_graphs_track_eager_steps():
    """five different B = 32 batches with look-ahead: eager twice (the run-to-run floor: the rasteriser's backward adds with atomics,
    which training amplifies) and once with the graphs on.  The first steps are eager in every run and must agree bit for bit; the
    capture step and the replays stay inside the floor; all three graphs really were captured."""
    a, b, c = _run(False), _run(False), _run(True)
    assert c["captured"] is True
    assert a["hist"][0] == b["hist"][0] == c["hist"][0]
    for x, y, z in zip(a["hist"], b["hist"], c["hist"]):
        for k in x:
            floor = abs(x[k] - y[k])
            assert abs(z[k] - x[k]) <= 6 * floor + 5e-3 * abs(x[k]) + 1e-9, (k, x[k], y[k], z[k])
