"""tools/graph_probe.py -- can the whole training step be captured in a HIP graph, and what does replay buy?
(timing probe: per-step host scalars -- jitter draw, rotation angle, learning rate -- are frozen at capture here)"""
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
# capturable optimizer for the probe
import scp_amd.optimizers as O  # noqa: E402
groups = tr.optim.optimizer.param_groups
tr.optim.optimizer = torch.optim.AdamW([{"params": g["params"], "lr": g["lr"]} for g in groups], lr=1e-4, betas=(0.9, 0.999),
                                       weight_decay=1e-4, fused=True, capturable=True)
tr.optim.scheduler = type("S", (), {"step": lambda self: None})()
torch.cuda.tunable.tuning_enable(False)
tr.model.rotation_angle = 90.0      # probe only: exact rot90 path, no per-step host->device angle matrix


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for _ in range(8):
    tr.step(data)
print("eager step %.2f ms" % timeit(lambda: tr.step(data)))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        tr.step(data)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = tr.step(data)
    print("captured")
    print("graph replay %.2f ms" % timeit(g.replay))
    print("loss after replays", float(out[0]))
except Exception as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("CAPTURE FAILED:", repr(e)[:400])
