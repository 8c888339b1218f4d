"""tools/conv_layers.py -- per-layer fp32 NHWC convolution times (MIOpen, benchmark mode) of the encoder at B=32, 256x256:
forward, input gradient, weight gradient, with the achieved TFLOP/s of each -- where the convolution milliseconds go."""
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = "cuda"
B = 32
# (name, Cin, Cout, k, stride, H_in, count per encoder pass)
LAYERS = [("stem 7x7/2", 3, 64, 7, 2, 256, 1), ("layer1 3x3", 64, 64, 3, 1, 64, 4), ("layer2.0 3x3/2", 64, 128, 3, 2, 64, 1),
          ("layer2 3x3", 128, 128, 3, 1, 32, 3), ("layer2 down 1x1/2", 64, 128, 1, 2, 64, 1), ("layer3.0 3x3/2", 128, 256, 3, 2, 32, 1),
          ("layer3 3x3", 256, 256, 3, 1, 16, 3), ("layer3 down 1x1/2", 128, 256, 1, 2, 32, 1), ("layer4.0 3x3/2", 256, 512, 3, 2, 16, 1),
          ("layer4 3x3", 512, 512, 3, 1, 8, 3), ("layer4 down 1x1/2", 256, 512, 1, 2, 16, 1),
          ("upconv5 3x3", 512, 256, 3, 1, 16, 1), ("iconv4 3x3", 512, 256, 3, 1, 16, 1), ("upconv4 3x3", 256, 128, 3, 1, 32, 1),
          ("iconv3 3x3", 256, 128, 3, 1, 32, 1), ("upconv3 3x3", 128, 64, 3, 1, 64, 1), ("iconv2 3x3", 128, 64, 3, 1, 64, 1),
          ("proj 1x1", 64, 64, 1, 1, 64, 1)]


def t(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0, 0.0]
print("%-20s %5s %8s | %8s %8s %8s us | TFLOP/s fwd dgrad wgrad" % ("layer", "count", "GFLOP", "fwd", "dgrad", "wgrad"))
for name, cin, cout, k, s, h, cnt in LAYERS:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(cout, cin, k, k, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(x, w, None, s, k // 2)
    g = torch.randn_like(y)
    ho = y.shape[2]
    gf = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
    tf = t(lambda: F.conv2d(x, w, None, s, k // 2))
    td = t(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    tw = t(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
    for i, v in enumerate((tf, td, tw)):
        tot[i] += cnt * v
    print("%-20s %5d %8.2f | %8.1f %8.1f %8.1f    | %5.0f %5.0f %5.0f" % (name, cnt, gf, tf, td, tw, gf / tf * 1e3, gf / td * 1e3, gf / tw * 1e3))
print("per encoder pass: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms (x2 passes per step)" % (tot[0] / 1e3, tot[1] / 1e3, tot[2] / 1e3))
