"""tools/conv_bench.py -- csrc/conv3x3.hip (implicit GEMM on the fp32 matrix cores, NHWC) vs MIOpen through F.conv2d on the
encoder's stride-1 3x3 layers at B=32: device time of the forward launch (profiler), agreement with a float64 convolution."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import capi  # noqa: E402

torch.backends.cudnn.benchmark = True
L = capi.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
ZEROS = torch.zeros(64, device="cuda")
B = int(os.environ.get("B", "32"))
LAYERS = [("layer1", 64, 64, 64), ("layer2", 128, 128, 32), ("layer3", 256, 256, 16), ("layer4", 512, 512, 8),
          ("upconv5", 512, 256, 16), ("upconv4", 256, 128, 32), ("upconv3", 128, 64, 64)]


def own(x_nhwc, w_khwc, bias, y_nhwc):
    n, h, w, cin = x_nhwc.shape
    capi.check(L.scp_conv3x3_nhwc_forward(P(x_nhwc), P(w_khwc), P(bias), P(ZEROS), P(y_nhwc), n, h, w, cin, w_khwc.shape[0],
                                          capi.current_stream()), "conv3x3")


def device_us(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return sum(e.self_device_time_total for e in prof.key_averages()) / n


for name, cin, cout, h in LAYERS:
    x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda")
    x_nhwc, w_khwc = x.permute(0, 2, 3, 1), w.permute(0, 2, 3, 1)          # views of the same storage
    assert x_nhwc.is_contiguous() and w_khwc.is_contiguous()
    y = torch.empty(B, h, h, cout, device="cuda")
    own(x_nhwc, w_khwc, b, y)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1).permute(0, 2, 3, 1)
    lib = F.conv2d(x, w, b, 1, 1).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    e_own, e_lib = (y.double() - ref).abs().max().item() / scale, (lib.double() - ref).abs().max().item() / scale
    t_own = device_us(lambda: own(x_nhwc, w_khwc, b, y))
    t_lib = device_us(lambda: F.conv2d(x, w, b, 1, 1))
    gf = 2.0 * B * h * h * cout * cin * 9 / 1e12       # TFLOP; / (us * 1e-6) -> TFLOP/s
    print("%-8s Cin %3d Cout %3d %3dx%-3d | own %7.1f us = %5.1f TF/s | MIOpen %7.1f us = %5.1f TF/s | err own %.1e MIOpen %.1e" % (
        name, cin, cout, h, h, t_own, gf / (t_own * 1e-6), t_lib, gf / (t_lib * 1e-6), e_own, e_lib))
