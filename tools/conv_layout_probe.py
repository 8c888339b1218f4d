"""tools/conv_layout_probe.py -- device time (profiler) and kernel names of the encoder's main 3x3 convolutions, fp32, in NCHW
vs NHWC: does MIOpen have a faster solver (e.g. Winograd) for the layout the encoder does NOT use?"""
import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

torch.backends.cudnn.benchmark = True
B = 32
LAYERS = [("layer1 3x3", 64, 64, 64), ("layer2 3x3", 128, 128, 32), ("layer3 3x3", 256, 256, 16), ("layer4 3x3", 512, 512, 8),
          ("iconv4 3x3", 512, 256, 16), ("iconv2 3x3", 128, 64, 64)]
for name, cin, cout, h in LAYERS:
    for fmt_name, fmt in (("NCHW", torch.contiguous_format), ("NHWC", torch.channels_last)):
        x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
        w = torch.randn(cout, cin, 3, 3, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
        g = torch.randn(B, cout, h, h, device="cuda").contiguous(memory_format=fmt)
        for _ in range(3):
            torch.autograd.grad(F.conv2d(x, w, None, 1, 1), (x, w), g)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                torch.autograd.grad(F.conv2d(x, w, None, 1, 1), (x, w), g)
            torch.cuda.synchronize()
        ev = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
        tot = sum(e.self_device_time_total for e in ev) / 5
        print("%-12s %s fwd+dgrad+wgrad device %.0f us: %s" % (name, fmt_name, tot, "; ".join("%s %.0f" % (e.key[:40], e.self_device_time_total / 5) for e in ev[:4])))
