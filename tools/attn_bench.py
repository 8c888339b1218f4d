"""tools/attn_bench.py -- HIP attention kernel vs torch SDPA at the DINO size (B=32, N=1025, 6x64)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd.dino import fused_attention  # noqa: E402

B, N, H = int(os.environ.get("B", "32")), int(os.environ.get("N", "1025")), 6
qkv = torch.randn(B, N, 3 * H * 64, device="cuda")


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def sdpa():
    q, k, v = qkv.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    return F.scaled_dot_product_attention(q, k, v, scale=0.125).transpose(1, 2).reshape(B, N, H * 64)


flop = 4.0 * B * H * N * N * 64
t1, t2 = timeit(lambda: fused_attention(qkv, B, N, H, 64, 0.125)), timeit(sdpa)
print("hip attention %.3f ms = %.1f TFLOP/s (%.1f%% of 157.3)   torch sdpa %.3f ms = %.1f TFLOP/s   max|diff| %.2e" % (
    t1, flop / t1 / 1e9, flop / t1 / 1e9 / 1.573, t2, flop / t2 / 1e9,
    (fused_attention(qkv, B, N, H, 64, 0.125) - sdpa()).abs().max().item()))

if "--ablate" in sys.argv:
    print("see tools/attn_probe.py (interleaved timing under the kernel's debug switches)")

if "--bf16" in sys.argv:
    from scp_amd.dino import fused_attention_bf16
    qb = qkv.to(torch.bfloat16)

    def sdpa_bf16():
        q, k, v = qb.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(q, k, v, scale=0.125).transpose(1, 2).reshape(B, N, H * 64)

    t3, t4 = timeit(lambda: fused_attention_bf16(qb, B, N, H, 64, 0.125)), timeit(sdpa_bf16)
    print("hip bf16 attention %.3f ms = %.1f TFLOP/s   torch sdpa bf16 %.3f ms = %.1f TFLOP/s   max|diff| %.2e" % (
        t3, flop / t3 / 1e9, t4, flop / t4 / 1e9, (fused_attention_bf16(qb, B, N, H, 64, 0.125).float() - sdpa_bf16().float()).abs().max().item()))
