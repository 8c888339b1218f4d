"""BatchNorm(+skip)(+ReLU) HIP op vs the stock torch composition evaluated in float64 on the CPU (the plain
PyTorch reference for this floating-point kernel), forward, running statistics and every gradient."""
import copy

import pytest
import torch
import torch.nn as nn

from scp_amd import fused_bn


def _case(n, c, h, w, skip, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g) * 1.7 + 0.4
    x = x + torch.linspace(-1, 1, c).view(1, c, 1, 1)            # channel-dependent mean
    s = torch.randn(n, c, h, w, generator=g) if skip else None
    bn = nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(c, generator=g))
        bn.bias.copy_(0.2 * torch.randn(c, generator=g))
        bn.running_mean.copy_(0.1 * torch.randn(c, generator=g))
        bn.running_var.copy_(1 + 0.2 * torch.rand(c, generator=g))
    dy = torch.randn(n, c, h, w, generator=g)
    return x, s, bn, dy


def _reference(x, s, bn, dy, relu, training):
    bn = copy.deepcopy(bn).double().train(training)
    x = x.double().requires_grad_(True)
    s = None if s is None else s.double().requires_grad_(True)
    y = fused_bn._composition(x, bn, s, relu)
    y.backward(dy.double())
    return y, x.grad, None if s is None else s.grad, bn.weight.grad, bn.bias.grad, bn


def test_cpu_path_is_the_composition():
    x, s, bn, _ = _case(2, 16, 4, 4, True, 0)
    ref = fused_bn._composition(x, copy.deepcopy(bn), s, True)
    assert torch.equal(fused_bn.bn_act(x, bn, s, relu=True), ref)


SHAPES = [(4, 64, 16, 16), (2, 128, 9, 7), (3, 512, 2, 2), (8, 16, 5, 5), (1, 256, 1, 1), (5, 64, 33, 31)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("skip,relu", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_matches_float64_composition(shape, skip, relu, training):
    if shape == (1, 256, 1, 1) and training:
        pytest.skip("one value per channel: torch refuses to train BatchNorm on it")
    x, s, bn, dy = _case(*shape, skip, seed=sum(shape))
    y_ref, dx_ref, ds_ref, dw_ref, db_ref, bn_ref = _reference(x, s, bn, dy, relu, training)
    dev = torch.device("cuda")
    bn_g = copy.deepcopy(bn).to(dev).train(training)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sg = None if s is None else s.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert fused_bn._fused_ok(xg, bn_g, sg)
    y = fused_bn.bn_act(xg, bn_g, sg, relu=relu)
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy.to(dev))

    def close(a, b, what, tol=2e-5):
        a, b = a.detach().double().cpu(), b.detach().double()
        err = (a - b).abs().max().item()
        assert err <= tol * (1 + b.abs().max().item()), "%s: %.3g" % (what, err)

    close(y, y_ref, "y")
    close(xg.grad, dx_ref, "dx")
    if skip:
        close(sg.grad, ds_ref, "dskip")
    close(bn_g.weight.grad, dw_ref, "dgamma", 5e-5)
    close(bn_g.bias.grad, db_ref, "dbeta", 5e-5)
    close(bn_g.running_mean, bn_ref.running_mean, "running_mean")
    close(bn_g.running_var, bn_ref.running_var, "running_var")
    assert int(bn_g.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.gpu
def test_bn_act_full_size_vs_miopen():
    """bench-size stem activation [32,64,128,128]: HIP op vs the MIOpen/ATen composition on the same device"""
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(32, 64, 128, 128, generator=g) * 2 + 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(32, 64, 128, 128, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    bn_a, bn_b = nn.BatchNorm2d(64).to(dev), nn.BatchNorm2d(64).to(dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = fused_bn.bn_act(xa, bn_a, relu=True)
    yb = fused_bn._composition(xb, bn_b, None, True)
    ya.backward(dy)
    yb.backward(dy)
    assert (ya - yb).abs().max().item() < 1e-5
    assert (xa.grad - xb.grad).abs().max().item() < 1e-5
    assert (bn_a.running_var - bn_b.running_var).abs().max().item() < 1e-6
    # frozen affine parameters (trainer.py:54-58) get no gradient buffers
    bn_a.weight.requires_grad_(False); bn_a.bias.requires_grad_(False)
    xa.grad = None
    fused_bn.bn_act(xa, bn_a, relu=True).backward(dy)
    assert bn_a.weight.grad is not None and xa.grad is not None   # old .grad kept, new dx produced


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 1, 1), (3, 16, 5, 7), (2, 128, 32, 32), (1, 4, 2, 9)])
def test_upsample2x_backward_matches_float64_autograd(shape):
    """decoder upsampling: HIP gather backward vs the float64 CPU autograd of F.interpolate (borders, odd sizes, 1x1)"""
    from scp_amd.nets import ResNet_Decoder, _Upsample2x
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    n, c, h, w = shape
    dy = torch.randn(n, c, 2 * h, 2 * w, generator=g)
    xr = x.double().requires_grad_(True)
    torch.nn.functional.interpolate(xr, (2 * h, 2 * w), mode="bilinear", align_corners=False).backward(dy.double())
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    like = torch.empty(n, c, 2 * h, 2 * w, device="cuda")
    y = ResNet_Decoder._up(xg, like)
    assert y.grad_fn is not None and "Upsample2x" in type(y.grad_fn).__name__
    y.backward(dy.cuda())
    assert (xg.grad.double().cpu() - xr.grad).abs().max().item() < 1e-5
    assert torch.allclose(y.detach().cpu().double(), torch.nn.functional.interpolate(x.double(), (2 * h, 2 * w), mode="bilinear", align_corners=False), atol=1e-6)


@pytest.mark.gpu
def test_upsample2x_backward_bf16():
    from scp_amd.nets import ResNet_Decoder
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 64, 9, 11, generator=g)
    dy = torch.randn(4, 64, 18, 22, generator=g)
    xr = x.double().requires_grad_(True)
    torch.nn.functional.interpolate(xr, (18, 22), mode="bilinear", align_corners=False).backward(dy.to(torch.bfloat16).double())
    xg = x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ResNet_Decoder._up(xg, torch.empty(4, 64, 18, 22, device="cuda"))
    assert "Upsample2x" in type(y.grad_fn).__name__ and y.dtype == torch.bfloat16
    y.backward(dy.cuda().to(torch.bfloat16))
    assert xg.grad.dtype == torch.bfloat16
    assert (xg.grad.double().cpu() - xr.grad).abs().max().item() < 2e-2 * (1 + xr.grad.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("skip,relu", [(False, True), (True, True), (False, False)])
def test_bn_act_bf16_storage(skip, relu):
    """configs[4] precision: bf16 activations in and out, fp32 statistics -- vs the float64 composition on the same
    (bf16-rounded) inputs; the band is bf16's 8 mantissa bits on outputs and gradients, statistics stay tight"""
    shape = (4, 64, 12, 10)
    x, s, bn, dy = _case(*shape, skip, seed=77)
    x, dy = x.to(torch.bfloat16).float(), dy.to(torch.bfloat16).float()
    s = None if s is None else s.to(torch.bfloat16).float()
    y_ref, dx_ref, ds_ref, _, _, bn_ref = _reference(x, s, bn, dy, relu, True)
    dev = torch.device("cuda")
    bn_g = copy.deepcopy(bn).to(dev).train(True)
    xg = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sg = None if s is None else s.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert fused_bn._fused_ok(xg, bn_g, sg)
    y = fused_bn.bn_act(xg, bn_g, sg, relu=relu)
    assert y.dtype == torch.bfloat16 and "BatchNormAct" in type(y.grad_fn).__name__
    y.backward(dy.to(dev).to(torch.bfloat16))

    def close(a, b, tol):
        a, b = a.detach().double().cpu(), b.detach().double()
        assert (a - b).abs().max().item() <= tol * (1 + b.abs().max().item())

    close(y, y_ref, 1e-2)
    close(xg.grad, dx_ref, 2e-2)
    if skip:
        close(sg.grad, ds_ref, 1e-2)
    close(bn_g.running_mean, bn_ref.running_mean, 1e-5)
    close(bn_g.running_var, bn_ref.running_var, 1e-5)


@pytest.mark.gpu
def test_bn_act_on_concurrent_streams_is_deterministic():
    """the last-workgroup fold of the statistics works through per-call ticket words: BatchNorm calls in flight on two
    streams at once (the two encoder passes of the training step) must neither disturb each other nor depend on timing"""
    from scp_amd.fused_bn import bn_act
    torch.manual_seed(0)
    shapes = [(8, 64, 32, 32), (8, 128, 16, 16), (4, 256, 8, 8), (16, 64, 64, 64)]
    xs = [torch.randn(s, device="cuda").contiguous(memory_format=torch.channels_last) for s in shapes]
    bns = [torch.nn.BatchNorm2d(s[1]).cuda().train() for s in shapes]

    def run(x, bn):
        xr = x.clone().requires_grad_(True)
        y = bn_act(xr, bn, None, True)
        (g,) = torch.autograd.grad(y.square().sum(), xr)
        return y.detach().clone(), g.clone()

    serial = [run(x, bn) for x, bn in zip(xs, bns)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(20):
        outs = [None] * len(xs)
        for i, (x, bn) in enumerate(zip(xs, bns)):
            st = streams[i % 2]
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs[i] = run(x, bn)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        for (y, g), (y0, g0) in zip(outs, serial):
            assert torch.equal(y, y0) and torch.equal(g, g0), rep


@pytest.mark.gpu
@pytest.mark.parametrize("shape,stride", [((4, 64, 16, 16), 1), ((2, 128, 9, 7), 1), ((3, 32, 8, 8), 2), ((32, 256, 16, 16), 1)])
def test_conv_bias_leaky_matches_the_stock_composition(shape, stride):
    """decoder conv unit: bias-free library convolution + fused bias/LeakyReLU (forward in place, one-pass backward with the
    bias gradient) against conv(bias) -> LeakyReLU evaluated in float64"""
    from scp_amd.fused_bn import conv_bias_leaky
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    conv = torch.nn.Conv2d(c, 64, 3, 1, 1).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(shape, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = conv_bias_leaky(x, conv, 0.1, stride)
    assert "BiasLeakyReLU" in type(y.grad_fn).__name__
    dy = torch.randn(y.shape, generator=g).cuda()
    gx, gw, gb = torch.autograd.grad(y, (x, conv.weight, conv.bias), dy)
    x64 = x.detach().double().requires_grad_(True)
    c64 = torch.nn.Conv2d(c, 64, 3, 1, 1).cuda().double()
    c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    pre = torch.nn.functional.conv2d(x64, c64.weight, c64.bias, stride, 1)
    assert (y.double() - torch.nn.functional.leaky_relu(pre, 0.1)).abs().max().item() <= 2e-5 * max(pre.abs().max().item(), 1.0)
    # gradients: the reference takes the fp32 run's branch of the activation (an output within rounding of zero flips between
    # the precisions and moves single gradient elements by O(1))
    ref = torch.where(y.detach() > 0, pre, 0.1 * pre)
    rx, rw, rb = torch.autograd.grad(ref, (x64, c64.weight, c64.bias), dy.double())
    for name, a, b in (("y", y, ref), ("dx", gx, rx), ("dw", gw, rw), ("db", gb, rb)):
        err = (a.double() - b).abs().max().item()
        assert err <= 2e-5 * max(b.abs().max().item(), 1.0), (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 8, 4, 6), (3, 64, 16, 16), (32, 64, 128, 128)])
def test_maxpool3x3s2_matches_aten_bitwise_including_ties(shape):
    """stem max pooling (csrc/pool.hip) vs F.max_pool2d on a ReLU output (a third of the values are exact zeros, so windows
    with several equal maxima are common): same values, and the gradient goes to the same element"""
    from scp_amd.fused_bn import maxpool3x3s2
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.relu(torch.randn(shape, generator=g)).cuda().contiguous(memory_format=torch.channels_last)
    stock = torch.nn.MaxPool2d(3, 2, 1)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = maxpool3x3s2(xa, stock)
    assert "MaxPool3x3s2" in type(ya.grad_fn).__name__
    yb = stock(xb)
    assert torch.equal(ya, yb)
    dy = torch.randn(ya.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    (ga,) = torch.autograd.grad(ya, xa, dy)
    (gb,) = torch.autograd.grad(yb, xb, dy)
    assert torch.equal(ga, gb)
