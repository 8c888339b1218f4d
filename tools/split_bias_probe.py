"""tools/split_bias_probe.py -- where does the split main loop's SYSTEMATIC error on same-sign data come from?
(tests/test_split_accuracy_gpu.py: BatchNorm sums from split conv outputs carry a relative bias of ~1e-7, the fp32 cores ~1e-8.)

All-positive operands.  For K in {32, 576, 4608}, signed mean and rms error against float64, in units of 2^-24 * mean|result|:
  repr      float64 sum of the six KEPT partial products (from the bf16 planes) vs the true product: what dropping ml, lm, ll costs
  bf16(hh)  the one-product instantiation (operands = the h planes, exactly representable): the matrix cores' own summation of
            positive bf16 products -- nothing is dropped, so any bias here is the hardware's accumulation
  fp32(hh)  the same h-plane operands through the fp32 matrix cores
  split     the shipped six-product kernel on the fp32 operands
  fp32      the fp32 matrix cores on the fp32 operands"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    sys.path.insert(0, p)
from scp_amd import dino, fused_conv  # noqa: E402

M, N = 2048, 384
g = torch.Generator().manual_seed(0)
print("%6s %-10s %12s %12s   (units of 2^-24 x mean |result|)" % ("K", "what", "signed mean", "rms"))
for K in (32, 576, 4608):
    a = (torch.rand(M, K, generator=g) + 0.5).cuda()
    w = ((torch.rand(N, K, generator=g) + 0.5) * 0.05).cuda()
    b = torch.zeros(N, device="cuda")
    true = a.double() @ w.double().t()
    unit = 2.0 ** -24 * true.abs().mean().item()
    pa, pw = fused_conv.split_planes(a).double(), fused_conv.split_planes(w).double()
    kept = sum(pa[i] @ pw[j].t() for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)))
    ah, wh = pa[0].float(), pw[0].float()
    hh = pa[0] @ pw[0].t()
    rows = [("repr", kept - true), ("bf16(hh)", dino.vit_linear(ah, wh, b, mode="bf16").double() - hh),
            ("fp32(hh)", dino.vit_linear(ah, wh, b, mode="fp32").double() - hh),
            ("split", dino.vit_linear(a, w, b, mode="split").double() - true),
            ("fp32", dino.vit_linear(a, w, b, mode="fp32").double() - true)]
    for name, d in rows:
        print("%6d %-10s %12.4f %12.4f" % (K, name, d.mean().item() / unit, d.square().mean().sqrt().item() / unit))
