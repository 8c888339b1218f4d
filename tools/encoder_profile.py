"""tools/encoder_profile.py -- the image encoder alone (first pass of the step: jitter+normalize, ResNet18 trunk, U-decoder,
heads), forward + backward at B=32; run under `rocprofv3 --kernel-trace --stats` to see where its non-convolution time goes."""
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
m = tr.model
B = data[0].shape[0]
mean_v = m.mesh.mean_v[None].expand(B, -1, -1)


def run():
    out = m.encoder(data[0], mean_v, data[9], data[7])
    loss = sum(o.float().square().mean() for o in out if torch.is_tensor(o) and o.requires_grad)
    loss.backward()
    for p_ in m.parameters():
        p_.grad = None


for _ in range(6):
    run()
torch.cuda.synchronize()
t = time.perf_counter()
N = 10
for _ in range(N):
    run()
torch.cuda.synchronize()
print("encoder forward+backward: %.2f ms" % ((time.perf_counter() - t) / N * 1e3))
