"""csrc/conv3x3.hip (3x3 / stride 1 / pad 1, NHWC fp32, implicit GEMM on the matrix cores) against a float64 convolution:
forward, and the input gradient obtained through the same entry point with transposed + flipped weights."""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))

pytestmark = pytest.mark.gpu


def _conv(x_nhwc, w_khwc, bias):
    from scp_amd import capi
    L = capi.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    n, h, w, cin = x_nhwc.shape
    y = torch.empty(n, h, w, w_khwc.shape[0], device="cuda")
    zeros = torch.zeros(16, device="cuda")
    capi.check(L.scp_conv3x3_nhwc_forward(P(x_nhwc), P(w_khwc), P(bias), P(zeros), P(y), n, h, w, cin, w_khwc.shape[0],
                                          capi.current_stream()), "conv3x3")
    return y


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 8, 8), (4, 32, 128, 16, 8), (2, 128, 200, 8, 16), (1, 16, 48, 16, 8),
                                           (32, 128, 128, 32, 32)])
def test_conv3x3_forward_and_input_gradient_vs_float64(n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
    b = torch.randn(cout, generator=g).cuda()
    x64 = x.double().requires_grad_(True)
    ref = F.conv2d(x64, wt.double(), b.double(), 1, 1)
    dy = torch.randn(n, cout, h, w, generator=g).cuda()
    (dx_ref,) = torch.autograd.grad(ref, x64, dy.double())
    y = _conv(x.permute(0, 2, 3, 1).contiguous(), wt.permute(0, 2, 3, 1).contiguous(), b)
    err = (y.double() - ref.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err
    if cout % 16 == 0:
        # dgrad = the same kernel: dy as input, weights [Cin, 3, 3, Cout] with the taps flipped, no bias
        w_t = wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()
        dx = _conv(dy.permute(0, 2, 3, 1).contiguous(), w_t, None)
        err = (dx.double() - dx_ref.permute(0, 2, 3, 1)).abs().max().item()
        assert err <= 1e-5 * dx_ref.abs().max().item(), err


def test_conv3x3_rejects_shapes_it_does_not_cover():
    from scp_amd import capi
    x = torch.randn(1, 4, 4, 24, device="cuda")      # Cin not a multiple of 16
    wt = torch.randn(8, 3, 3, 24, device="cuda")
    y = torch.empty(1, 4, 4, 8, device="cuda")
    z = torch.zeros(16, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert capi.lib().scp_conv3x3_nhwc_forward(P(x), P(wt), ctypes.c_void_p(0), P(z), P(y), 1, 4, 4, 24, 8, capi.current_stream()) != 0
