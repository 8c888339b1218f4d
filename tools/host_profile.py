"""tools/host_profile.py -- cProfile of the host side of the bench workload's step (no synchronisation inside the profiled loop):
where the ~34 ms of Python / dispatcher time per step go."""
import cProfile
import os
import pstats
import sys

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
torch.cuda.synchronize()
N = 10
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.step(data)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(70)
st.sort_stats("tottime").print_stats(45)
