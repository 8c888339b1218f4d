"""tools/probes/dino_prune_probe.py -- DINO key features at B=32, 256x256: full pass vs the pass that runs the last block's
proj / MLP and the key projection on the masked tokens only (scp_amd/dino.py tail_keys_fused)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import dino, synthetic  # noqa: E402

dino.ALLOW_RANDOM_INIT = True
net = dino.DINO("/nonexistent.pth").cuda()
data = synthetic.make_batch(8, 4, 256, seed=100, device="cuda")
img, mask = data[0], data[1]
keep = torch.nn.functional.interpolate(mask[:, None].float(), (32, 32), mode="nearest").reshape(32, -1) > 0
print("kept token fraction %.3f" % keep.float().mean().item())


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("full   %.3f ms" % t(lambda: net(img)))
print("pruned %.3f ms" % t(lambda: net(img, keep)))

from torch.profiler import ProfilerActivity, profile  # noqa: E402
for tag, fn in (("full", lambda: net(img)), ("pruned", lambda: net(img, keep))):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    ev = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
    print(tag, "device total %.2f ms:" % (sum(e.self_device_time_total for e in ev) / 5e3),
          "; ".join("%s x%d %.0f" % (e.key[:34], e.count // 5, e.self_device_time_total / 5) for e in ev[:14]))
