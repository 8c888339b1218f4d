"""tools/probes/main_priority_probe.py -- step time with the MAIN stream at high priority (side streams normal), and with the ViT
side stream at low priority; tools/priority_probe.py covers the side streams at high priority (+23 ms)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402
from scp_amd import synthetic  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
data = synthetic.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")


def timed(n=30, w=6):
    for _ in range(w):
        tr.step(data)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("default: %.2f ms" % timed())
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
hs = torch.cuda.Stream(priority=-1)
hs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(hs):
    print("main stream high priority: %.2f ms" % timed())
torch.cuda.current_stream().wait_stream(hs)
print("default again: %.2f ms" % timed())
