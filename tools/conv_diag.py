"""tools/conv_diag.py -- times the stock-PyTorch encoder (ResNet18 + decoder) fwd+bwd at B=32 under
different MIOpen / layout settings, and lists the slowest kernels (diagnostic for naive-conv fallbacks)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from scp_amd.flags import Options  # noqa: E402
from scp_amd.encoder import Encoder  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "default"
torch.backends.cudnn.benchmark = "bench" in variant
opts = Options("laptop_wild6d", batch_size=8, repeat=4, train=True, mixed_bf16="bf16" in variant)
torch.manual_seed(0)
enc = Encoder(opts).cuda().train()
NB = int(os.environ.get("NB", "32"))
x = torch.rand(NB, 3, 256, 256, device="cuda")
if "cl" in variant:
    enc = enc.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)


def step():
    code, feat = enc.encode_img(x)          # "bf16" variants: opts.mixed_bf16 (the encoder applies its own autocast)
    (feat.float().square().mean() + code.float().mean()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("%-16s encoder fwd+bwd B=%d: %.2f ms" % (variant, NB, (time.perf_counter() - t) / 5 * 1e3))
if "prof" in variant:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
