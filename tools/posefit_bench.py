"""tools/posefit_bench.py -- test-time pose fitting throughput (images/s) at the bench geometry: B=32 images of
256x256 with ~20k valid correspondences each, 100 RANSAC rounds per image; HIP batched path vs the numpy oracle
(a float32 restatement of the reference's per-image torch loop) on a sample of the same batch."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from posefit_inputs import posefit_inputs  # noqa: E402
from scp_amd import pose_fit  # noqa: E402

B = int(os.environ.get("B", "32"))
data, _ = posefit_inputs(bsz=B, size=256, n_verts=642, seed=3)
keys = ("depth", "mask", "match", "match_conf", "foc_crop", "pp_crop", "pred_v")
dev_in = [data[k].cuda() for k in keys]
fit = pose_fit.PoseFitter(img_size=256, base_rot=torch.eye(3)[None].cuda())
for _ in range(3):
    fit.pose_fitting(*dev_in)
torch.cuda.synchronize()
n = 20
t = time.perf_counter()
for _ in range(n):
    fit.pose_fitting(*dev_in)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t) / n * 1e3
pts = pose_fit.last_report["n_points"]

from oracle import posefit as oracle_posefit  # noqa: E402  (CPU baseline leg only)
sample = min(B, 4)
host_in = [data[k][:sample].numpy() for k in keys]
t = time.perf_counter()
oracle_posefit.pose_fitting_oracle(*host_in, torch.eye(3)[None].numpy(), lambda m: torch.randint(0, m, (5,)).numpy())
cpu_s = time.perf_counter() - t
print(json.dumps({"metric": "pose fits/sec (256x256, 100 RANSAC rounds, ~%dk correspondences)" % (sum(pts) // len(pts) // 1000),
                  "value": B / gpu_ms * 1e3, "ms_per_batch": gpu_ms, "batch": B,
                  "cpu_baseline": {"value": sample / cpu_s, "kind": "port", "cores": 1, "sample": "%d images" % sample}}))
