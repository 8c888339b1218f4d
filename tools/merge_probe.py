"""tools/merge_probe.py -- timing probe (NOT numerically equivalent: BatchNorm statistics are taken over both halves): what would
the step cost if the two encoder passes (images, rotated images) ran as ONE B=64 pass?  Decides whether the per-half-statistics
version is worth building."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402
from scp_amd import imgops  # noqa: E402
from scp_amd.encoder import Encoder  # noqa: E402


def timed(tr, data, n=30, w=8):
    for _ in range(w):
        tr.step(data)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        tr.step(data)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


tr, opts = bench.build_trainer("cuda", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
tr.model.rotation_angle = 37.0
print("baseline (two passes, rotation cycle on its side stream): %.2f ms" % timed(tr, data))

orig_encode = Encoder.encode_img
orig_forward = Encoder.forward


def merged_forward(self, img, mean_v, pp_crop, foc_crop):
    b = img.shape[0]
    tgt = imgops.rotate(img, 37.0, "bilinear")
    code, feat = orig_encode(self, torch.cat((img, tgt)))
    self._stash = (code[b:], feat[b:])
    self._first = (code[:b], feat[:b])
    Encoder.encode_img = lambda s, x: s._first
    try:
        out = orig_forward(self, img, mean_v, pp_crop, foc_crop)
    finally:
        Encoder.encode_img = lambda s, x: s._stash
    return out


Encoder.forward = merged_forward
for side in (True, False):
    tr.model.overlap_rotation_cycle = side
    print("merged B=64 pass, rotation-cycle remainder on %s stream: %.2f ms" % ("its side" if side else "the main", timed(tr, data)))
