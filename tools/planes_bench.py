"""tools/planes_bench.py -- the batched weight-plane refresh (fused_conv.refresh_planes -> scp_conv_weight_planes_batch) of the
encoder's real layer list against one scp_conv_weight_planes launch per layer: time per refresh (HIP events, 50 repetitions each) and
bit equality of every plane set.  Run on the GPU box: python tools/planes_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402
from scp_amd import fused_conv  # noqa: E402

tr, opts = bench.build_trainer("cuda:0", 1)
convs = tr._convs


def refresh(batch):
    os.environ["SCP_PLANES_BATCH"] = batch
    fused_conv.invalidate()
    fused_conv.refresh_planes(convs)


def timed(batch, reps=50):
    for _ in range(3):
        refresh(batch)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        refresh(batch)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e3


snaps = {}
for batch in ("0", "1"):
    refresh(batch)
    torch.cuda.synchronize()
    snaps[batch] = [{k: c._scp_planes[k].clone() for k in ("fwd", "dgrad") if k in c.__dict__.get("_scp_planes", {})} for c in convs]
same = all(a.keys() == b.keys() and all(torch.equal(a[k].view(torch.int16), b[k].view(torch.int16)) for k in a)
           for a, b in zip(snaps["0"], snaps["1"]))
nbytes = sum(c.weight.numel() for c in convs if "_scp_planes" in c.__dict__) * 4
print("layers with planes: %d, weights %.1f MB, plane sets bit-equal: %s" % (sum("_scp_planes" in c.__dict__ for c in convs), nbytes / 1e6, same))
print("per-layer launches: %.1f us per refresh (host loop included)" % timed("0"))
print("one batched launch: %.1f us per refresh (host loop included)" % timed("1"))
