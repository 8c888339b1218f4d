"""csrc/conv_igemm.hip / csrc/conv_wgrad.hip (3x3 and 1x1 convolutions of the image encoder, NHWC fp32, implicit GEMM on the fp32
matrix cores) against float64 convolutions: forward (stride 1 and 2, raw / bias + LeakyReLU epilogues, per-tile BatchNorm
partial sums), the input gradient through the same entry point with transposed + flipped weights, and the weight gradient.
Tolerance 1e-5 of the output scale (fp32 accumulation over K <= 4608 in MFMA order)."""
import ctypes
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))

pytestmark = pytest.mark.gpu
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


CORE = "split"      # which main loop _conv() drives: bf16 cores on exactly split operands, or the fp32 cores


@pytest.fixture(params=["split", "fp32"], autouse=True)
def conv_core(request):
    global CORE
    CORE = request.param
    yield request.param
    CORE = "split"


def _conv(x_nhwc, w_khwc, bias=None, stride=1, leaky=False, slope=0.1, partials=False):
    from scp_amd import capi, fused_conv
    L = capi.lib()
    w3 = fused_conv.split_planes_tiled(w_khwc) if CORE == "split" else None
    n, h, w, cin = x_nhwc.shape
    cout, k = w_khwc.shape[0], w_khwc.shape[1]
    ho = (h + 2 * (k // 2) - k) // stride + 1
    wo = (w + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(n, ho, wo, cout, device="cuda")
    part = None
    if partials:
        tm, rows = ctypes.c_int(), ctypes.c_int()
        L.scp_conv_nhwc_partial_rows(n, h, w, cin, cout, k, stride, int(CORE == "split"), ctypes.byref(tm), ctypes.byref(rows))
        part = torch.full((2, tm.value, cout), float("nan"), device="cuda")
    sk_bytes = L.scp_conv_nhwc_splitk_workspace(n, h, w, cin, cout, k, stride, int(CORE == "split"))
    sk = torch.empty(sk_bytes // 4, device="cuda") if sk_bytes else None
    capi.check(L.scp_conv_nhwc_forward(P(x_nhwc), P(None if w3 is not None else w_khwc), P(w3), P(bias), P(y), P(part), n, h, w, cin, cout,
                                       k, stride, int(leaky), slope, P(sk), sk_bytes, capi.current_stream()), "conv_nhwc_forward")
    return (y, part, rows.value) if partials else y


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride", [
    (2, 64, 64, 8, 8, 3, 1), (4, 32, 128, 16, 8, 3, 1), (2, 128, 200, 8, 16, 3, 1), (3, 64, 48, 16, 8, 3, 1),
    (2, 64, 128, 16, 16, 3, 2), (2, 128, 256, 8, 8, 3, 2), (2, 64, 128, 16, 16, 1, 2), (2, 64, 64, 16, 16, 1, 1),
    (32, 128, 128, 32, 32, 3, 1), (8, 64, 64, 64, 64, 3, 1), (4, 512, 512, 8, 8, 3, 1), (1, 256, 512, 4, 4, 3, 1),
    # the split main loop runs these split-K (2 / 4 / 8 workgroups per 128 x 128 tile + fold pass):
    (32, 256, 256, 16, 16, 3, 1), (32, 512, 512, 8, 8, 3, 1), (32, 128, 256, 32, 32, 3, 2), (9, 512, 256, 16, 16, 3, 1)])
def test_conv_forward_and_input_gradient_vs_float64(n, cin, cout, h, w, k, stride):
    g = torch.Generator().manual_seed(cin + cout + h + k + stride)
    x = torch.randn(n, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.1).cuda()
    x64 = x.double().requires_grad_(True)
    ref = F.conv2d(x64, wt.double(), None, stride, k // 2)
    y = _conv(x.permute(0, 2, 3, 1).contiguous(), wt.permute(0, 2, 3, 1).contiguous(), stride=stride)
    err = (y.double() - ref.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err
    if stride == 1 and cout % 32 == 0 and (cout & (cout - 1)) == 0:
        # dgrad = the same kernel: dy as input, weights [Cin, k, k, Cout] with the taps flipped
        dy = torch.randn(n, cout, h, w, generator=g).cuda()
        (dx_ref,) = torch.autograd.grad(ref, x64, dy.double())
        w_t = wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()
        dx = _conv(dy.permute(0, 2, 3, 1).contiguous(), w_t)
        err = (dx.double() - dx_ref.permute(0, 2, 3, 1)).abs().max().item()
        assert err <= 1e-5 * dx_ref.abs().max().item(), err


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 128, 16, 16), (3, 32, 64, 8, 12), (32, 64, 128, 64, 64), (32, 128, 256, 32, 32),
                                           (32, 256, 512, 16, 16), (5, 128, 128, 6, 10), (1, 192, 64, 4, 4)])
def test_stride2_input_gradient_vs_float64(n, cin, cout, h, w, conv_core):
    """scp_conv_nhwc_dgrad_stride2: the 3x3 / stride-2 / pad-1 input gradient by parity class (split main loop only), every dx element
    written exactly once (the output starts as NaN)"""
    if conv_core != "split":
        pytest.skip("the stride-2 input gradient runs on the split main loop only")
    from scp_amd import capi, fused_conv
    L = capi.lib()
    g = torch.Generator().manual_seed(n + cin + cout + h)
    x64 = torch.randn(n, cin, h, w, generator=g).double().cuda().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
    ref = F.conv2d(x64, wt.double(), None, 2, 1)
    dy = torch.randn(ref.shape, generator=g).cuda()
    (dx_ref,) = torch.autograd.grad(ref, x64, dy.double())
    fwd = torch.empty(fused_conv.tiled_planes_numel(cout, 9 * cin), dtype=torch.bfloat16, device="cuda")
    dgr = torch.empty(fused_conv.tiled_planes_numel(cin, 9 * cout), dtype=torch.bfloat16, device="cuda")
    capi.check(L.scp_conv_weight_planes(P(wt), wt.stride(0), wt.stride(1), wt.stride(2), wt.stride(3), cout, cin, 3, P(fwd), P(dgr),
                                        capi.current_stream()), "conv_weight_planes")
    dx = torch.full((n, h, w, cin), float("nan"), device="cuda")
    capi.check(L.scp_conv_nhwc_dgrad_stride2(P(dy.permute(0, 2, 3, 1).contiguous()), P(dgr), P(dx), n, h // 2, w // 2, cout, cin,
                                             capi.current_stream()), "conv_nhwc_dgrad_stride2")
    err = (dx.double() - dx_ref.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= 1e-5 * dx_ref.abs().max().item(), err


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 16, 16), (3, 128, 128, 8, 8), (2, 64, 256, 8, 8), (32, 64, 64, 64, 64),
                                           (32, 256, 256, 16, 16), (31, 512, 512, 8, 8)])   # the last two: split-K + fold epilogue
def test_conv_epilogues(n, cin, cout, h, w):
    """bias + LeakyReLU(0.1) of the decoder's conv units, and the per-tile column sums a BatchNorm folds into its statistics
    (always of the RAW convolution output)"""
    g = torch.Generator().manual_seed(n + cin + cout)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.1).cuda()
    b = torch.randn(cout, generator=g).cuda()
    raw64 = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), None, 1, 1).permute(0, 2, 3, 1)
    ref = F.leaky_relu(raw64 + b.double(), 0.1)
    y = _conv(x, wt, b, leaky=True)
    assert (y.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    y, part, rows = _conv(x, wt, None, partials=True)
    assert (y.double() - raw64).abs().max().item() <= 1e-5 * raw64.abs().max().item()
    flat = raw64.reshape(-1, cout)
    pad = (-flat.shape[0]) % rows
    tiles = torch.cat((flat, flat.new_zeros(pad, cout))).reshape(-1, rows, cout)
    assert part.shape[1] == tiles.shape[0] and torch.isfinite(part).all()
    torch.testing.assert_close(part[0].double(), tiles.sum(1), rtol=1e-4, atol=1e-4 * float(flat.abs().max()) * rows ** 0.5)
    torch.testing.assert_close(part[1].double(), (tiles * tiles).sum(1), rtol=1e-4, atol=1e-4 * float(flat.abs().max()) ** 2 * rows ** 0.5)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 16, 16), (2, 128, 64, 32, 16), (4, 64, 128, 8, 8), (3, 128, 128, 8, 8),
                                           (32, 64, 64, 64, 64), (32, 512, 512, 8, 8), (32, 256, 128, 32, 32)])
def test_conv_weight_gradient_vs_float64(n, cin, cout, h, w):
    from scp_amd import capi
    L = capi.lib()
    g = torch.Generator().manual_seed(cin * 3 + cout + h)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    w64 = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    out = F.conv2d(x.permute(0, 3, 1, 2).double(), w64, None, 1, 1)
    (dw_ref,) = torch.autograd.grad(out, w64, dy.permute(0, 3, 1, 2).double())
    ws_bytes = L.scp_conv_nhwc_weight_grad_workspace(n, h, w, cin, cout, 3, 1)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes // 4, device="cuda")
    dw = torch.full((cout, 3, 3, cin), float("nan"), device="cuda")
    for _ in range(2):           # twice: the second call must not depend on what the first left in the workspace
        capi.check(L.scp_conv_nhwc_weight_grad(P(x), P(dy), P(dw), P(None), P(ws), ws_bytes, n, h, w, cin, cout, 3, 1,
                                               int(CORE == "split"), capi.current_stream()), "conv_nhwc_weight_grad")
    err = (dw.double() - dw_ref.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= 2e-5 * dw_ref.abs().max().item(), (err, dw_ref.abs().max().item())


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 16, 16), (32, 64, 64, 64, 64), (3, 128, 64, 8, 32)])
def test_1x1_weight_gradient_vs_float64(n, cin, cout, h, w, conv_core):
    """1x1 / stride 1 (the decoder's feature projection) on the halo-block kernel with a one-pixel block (split core)"""
    if conv_core != "split":
        pytest.skip("split core only")
    from scp_amd import capi
    L = capi.lib()
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    dy = torch.randn(n, h, w, cout, generator=g).cuda()
    dw_ref = torch.einsum("nhwo,nhwi->oi", dy.double(), x.double())
    ws_bytes = L.scp_conv_nhwc_weight_grad_workspace(n, h, w, cin, cout, 1, 1)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes // 4, device="cuda")
    dw = torch.full((cout, 1, 1, cin), float("nan"), device="cuda")
    capi.check(L.scp_conv_nhwc_weight_grad(P(x), P(dy), P(dw), P(None), P(ws), ws_bytes, n, h, w, cin, cout, 1, 1, 1, capi.current_stream()), "wgrad 1x1")
    err = (dw.reshape(cout, cin).double() - dw_ref).abs().max().item()
    assert err <= 2e-5 * dw_ref.abs().max().item(), (err, dw_ref.abs().max().item())


@pytest.mark.parametrize("n,cin,cout,h,w,k", [(2, 64, 128, 32, 32, 3), (2, 64, 64, 16, 32, 3), (32, 64, 128, 64, 64, 3), (32, 128, 256, 32, 32, 3),
                                             (32, 256, 512, 16, 16, 3), (4, 64, 128, 16, 16, 1), (32, 64, 128, 64, 64, 1), (32, 256, 512, 16, 16, 1),
                                             (2, 128, 64, 32, 64, 1)])
def test_stride2_weight_gradient_vs_float64(n, cin, cout, h, w, k, conv_core):
    """the stride-2 layers of the trunk (3x3 pad 1 and the 1x1 projections) on the halo-block kernel with a strided block"""
    if conv_core != "split":
        pytest.skip("stride-2 weight gradients run on the split core only")
    from scp_amd import capi
    L = capi.lib()
    g = torch.Generator().manual_seed(cin * 5 + cout + h + k)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    dy = torch.randn(n, h // 2, w // 2, cout, generator=g).cuda()
    w64 = torch.zeros(cout, cin, k, k, dtype=torch.float64, device="cuda", requires_grad=True)
    out = F.conv2d(x.permute(0, 3, 1, 2).double(), w64, None, 2, k // 2)
    (dw_ref,) = torch.autograd.grad(out, w64, dy.permute(0, 3, 1, 2).double())
    ws_bytes = L.scp_conv_nhwc_weight_grad_workspace(n, h, w, cin, cout, k, 2)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes // 4, device="cuda")
    dw = torch.full((cout, k, k, cin), float("nan"), device="cuda")
    for _ in range(2):
        capi.check(L.scp_conv_nhwc_weight_grad(P(x), P(dy), P(dw), P(None), P(ws), ws_bytes, n, h, w, cin, cout, k, 2, 1,
                                               capi.current_stream()), "conv_nhwc_weight_grad")
    err = (dw.double() - dw_ref.permute(0, 2, 3, 1)).abs().max().item()
    assert err <= 2e-5 * dw_ref.abs().max().item(), (err, dw_ref.abs().max().item())


def test_conv_rejects_shapes_it_does_not_cover():
    from scp_amd import capi
    L = capi.lib()
    x = torch.randn(1, 4, 4, 24, device="cuda")      # Cin not a power of two >= 32
    wt = torch.randn(8, 3, 3, 24, device="cuda")
    y = torch.empty(1, 4, 4, 8, device="cuda")
    assert L.scp_conv_nhwc_forward(P(x), P(wt), P(None), P(None), P(y), P(None), 1, 4, 4, 24, 8, 3, 1, 0, 0.0, P(None), 0, capi.current_stream()) != 0
    assert L.scp_conv_nhwc_forward(P(x), P(wt), P(None), P(None), P(y), P(None), 1, 4, 4, 32, 8, 7, 2, 0, 0.0, P(None), 0, capi.current_stream()) != 0   # the 7x7 stem stays on MIOpen
    assert L.scp_conv_nhwc_weight_grad_workspace(2, 12, 12, 64, 64, 3, 1) == 0                                                 # not a power-of-two map
