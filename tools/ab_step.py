"""tools/ab_step.py -- A/B of model-level switches inside ONE process on the GPU box: step time of the bench workload
(B=32, 256x256) with each listed attribute of MeshNet toggled.  Usage: python tools/ab_step.py eager_cycle_backward [more attrs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402
from scp_amd import synthetic as synth  # noqa: E402

tr, opts = bench.build_trainer("cuda:0", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda:0")
for _ in range(6):
    tr.step(data)


def timed(n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for attr in sys.argv[1:]:
    for rep in range(2):
        for val in (True, False):
            setattr(tr.model, attr, val)
            for _ in range(3):
                tr.step(data)
            print("%s=%s: %.2f ms/step" % (attr, val, timed()), flush=True)

if os.environ.get("AB_PRIORITY") == "1":
    hs = torch.cuda.Stream(priority=-1)
    hs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hs):
        for val in (True, False, True, False):
            tr.model.eager_cycle_backward = val
            for _ in range(3):
                tr.step(data)
            print("main stream HIGH priority, eager_cycle_backward=%s: %.2f ms/step" % (val, timed()), flush=True)
    torch.cuda.current_stream().wait_stream(hs)
