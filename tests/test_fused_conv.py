"""scp_amd/fused_conv.py on the GPU: convolution + BatchNorm (+ skip) (+ ReLU) and convolution + bias + LeakyReLU as single
autograd ops on the own implicit-GEMM kernels, against the stock torch composition in float64 (values, input / weight / affine
gradients, running statistics), and the whole image encoder against itself with the fused path switched off."""
import os
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["split", "fp32"], autouse=True)
def conv_mode(request):
    """both matrix-core paths of the own convolutions (scp_amd/fused_conv.py CONV_MODE)"""
    from scp_amd import fused_conv
    old, fused_conv.CONV_MODE = fused_conv.CONV_MODE, request.param
    yield request.param
    fused_conv.CONV_MODE = old


def _close(got, ref, tol, what):
    err = (got.double() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "%s: max err %.3e of scale %.3e" % (what, err, scale)


def _close_grad(got, ref, tol, what):
    """gradients behind a ReLU: an output within rounding of zero takes the other branch in float64 and moves a handful of
    elements by O(1) -- judged in relative L2 and by the fraction of elements beyond the max-abs tolerance"""
    d = (got.double() - ref).abs()
    scale = max(ref.abs().max().item(), 1e-6)
    rel = (d.square().sum().sqrt() / ref.square().sum().sqrt().clamp_min(1e-30)).item()
    frac = (d > tol * scale).double().mean().item()
    assert rel <= 4 * tol and frac <= 1e-4, "%s: rel L2 %.3e, %.2e of the elements beyond %.1e of scale" % (what, rel, frac, tol)


@pytest.mark.parametrize("n,cin,cout,h,k,stride,skip,relu", [
    (4, 64, 64, 16, 3, 1, False, True), (4, 64, 64, 16, 3, 1, True, True), (2, 64, 128, 16, 3, 2, False, True),
    (2, 64, 128, 16, 1, 2, False, False), (3, 128, 256, 16, 1, 2, False, False), (2, 128, 128, 8, 3, 1, True, True), (2, 256, 512, 8, 3, 2, False, True),
    (32, 64, 64, 64, 3, 1, True, True)])
def test_conv_bn_act_vs_float64(n, cin, cout, h, k, stride, skip, relu):
    from scp_amd import fused_conv
    g = torch.Generator().manual_seed(n + cin + cout + k)
    conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(cout).cuda()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(cout, generator=g))
        bn.bias.copy_(0.3 * torch.randn(cout, generator=g))
    x = torch.randn(n, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    sk = torch.randn(n, cout, ho, ho, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if skip else None
    dy = torch.randn(n, cout, ho, ho, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    # float64 reference
    conv64, bn64 = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).cuda().double(), nn.BatchNorm2d(cout).cuda().double()
    conv64.weight.data.copy_(conv.weight.double())
    bn64.weight.data.copy_(bn.weight.double()); bn64.bias.data.copy_(bn.bias.double())
    x64 = x.detach().double().requires_grad_(True)
    sk64 = sk.detach().double().requires_grad_(True) if skip else None
    y = fused_conv.conv_bn_act(x, conv, bn, skip=sk, relu=relu)
    assert y.grad_fn is not None and "ConvBNAct" in type(y.grad_fn).__name__, "the fused op must be the one that runs"
    y.backward(dy)
    r = bn64(conv64(x64))
    if skip:
        r = r + sk64
    if relu:
        _close(y, F.relu(r), 2e-5, "output")
        # the reference's ReLU takes the fp32 run's branch: an output within rounding of zero would otherwise flip between the
        # two precisions and move a handful of gradient elements by O(1), which says nothing about the kernels
        r = r * (y.detach() > 0).double()
    r.backward(dy.double())
    _close(y, r, 2e-5, "output")
    _close_grad(x.grad, x64.grad, 5e-5, "dx")
    _close_grad(conv.weight.grad, conv64.weight.grad, 5e-5, "dw")
    _close_grad(bn.weight.grad, bn64.weight.grad, 5e-5, "dgamma")
    _close_grad(bn.bias.grad, bn64.bias.grad, 5e-5, "dbeta")
    if skip:
        _close_grad(sk.grad, sk64.grad, 5e-5, "dskip")
    _close(bn.running_mean, bn64.running_mean, 1e-5, "running_mean")
    _close(bn.running_var, bn64.running_var, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("n,h,w,channels_last", [(2, 32, 32, False), (3, 64, 48, True), (2, 256, 256, False), (32, 256, 256, False),
                                                (1, 512, 512, False), (2, 40, 264, False)])
def test_stem_conv_bn_act_vs_float64(n, h, w, channels_last):
    """the 7x7 / stride-2 stem + BatchNorm + ReLU on csrc/conv_stem.hip: output, running statistics, weight / gamma / beta gradients"""
    from scp_amd import fused_conv
    g = torch.Generator().manual_seed(n + h + w)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    if channels_last:
        conv = conv.to(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(64).cuda()
    with torch.no_grad():
        bn.weight.copy_(1 + 0.3 * torch.randn(64, generator=g))
        bn.bias.copy_(0.3 * torch.randn(64, generator=g))
    x = torch.rand(n, 3, h, w, generator=g).cuda()          # an image: all positive
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, 64, h // 2, w // 2, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    conv64, bn64 = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda().double(), nn.BatchNorm2d(64).cuda().double()
    conv64.weight.data.copy_(conv.weight.double())
    bn64.weight.data.copy_(bn.weight.double()); bn64.bias.data.copy_(bn.bias.double())
    y = fused_conv.stem_conv_bn_act(x, conv, bn, relu=True)
    assert y.grad_fn is not None and "StemConvBNAct" in type(y.grad_fn).__name__, "the fused op must be the one that runs"
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    r = bn64(conv64(x.double()))
    _close(y, F.relu(r), 2e-5, "output")
    r = r * (y.detach() > 0).double()
    r.backward(dy.double())
    _close_grad(conv.weight.grad, conv64.weight.grad, 5e-5, "dw")
    _close_grad(bn.weight.grad, bn64.weight.grad, 5e-5, "dgamma")
    _close_grad(bn.bias.grad, bn64.bias.grad, 5e-5, "dbeta")
    _close(bn.running_mean, bn64.running_mean, 1e-5, "running_mean")
    _close(bn.running_var, bn64.running_var, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == 1
    # deterministic: a second run reproduces the first bit for bit
    conv.weight.grad = None
    y2 = fused_conv.stem_conv_bn_act(x, conv, bn, relu=True)
    y2.backward(dy)
    assert torch.equal(y2, y)
    g1 = conv.weight.grad.clone()
    conv.weight.grad = None
    fused_conv.stem_conv_bn_act(x, conv, bn, relu=True).backward(dy)
    assert torch.equal(conv.weight.grad, g1)


@pytest.mark.parametrize("n,cin,cout,h,stride", [(2, 128, 64, 16, 1), (2, 512, 256, 8, 1), (2, 128, 64, 16, 2), (32, 128, 64, 64, 1)])
def test_conv_bias_leaky_vs_float64(n, cin, cout, h, stride):
    from scp_amd import fused_conv
    g = torch.Generator().manual_seed(n + cin + cout + stride)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(n, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x64 = x.detach().double().requires_grad_(True)
    w64, b64 = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    y = fused_conv.conv_bias_leaky(x, conv, 0.1, stride)
    assert "ConvBiasLeaky" in type(y.grad_fn).__name__
    pre = F.conv2d(x64, w64, b64, stride, 1)
    _close(y, F.leaky_relu(pre, 0.1), 2e-5, "output")
    r = torch.where(y.detach() > 0, pre, 0.1 * pre)          # the fp32 run's branch (see test_conv_bn_act_vs_float64)
    dy = torch.randn(r.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    r.backward(dy.double())
    y.backward(dy)
    _close_grad(x.grad, x64.grad, 5e-5, "dx")
    _close_grad(conv.weight.grad, w64.grad, 5e-5, "dw")
    _close_grad(conv.bias.grad, b64.grad, 5e-5, "dbias")


@pytest.mark.parametrize("n,cin,cout,h", [(2, 64, 64, 16), (32, 64, 64, 64), (3, 128, 16, 32)])
def test_projection_1x1_with_bias_vs_float64(n, cin, cout, h):
    """the decoder's 1x1 feature projection (ResNet_Decoder._project): conv + bias through the LeakyReLU epilogue with slope 1 --
    the identity -- forward and all three gradients"""
    from scp_amd import fused_conv
    g = torch.Generator().manual_seed(n + cin + cout)
    conv = nn.Conv2d(cin, cout, 1, bias=True).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(n, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x64 = x.detach().double().requires_grad_(True)
    w64, b64 = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    y = fused_conv.conv_bias_leaky(x, conv, 1.0)
    assert "ConvBiasLeaky" in type(y.grad_fn).__name__
    r = F.conv2d(x64, w64, b64)
    _close(y, r, 2e-5, "output")
    dy = torch.randn(r.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    r.backward(dy.double())
    y.backward(dy)
    _close(x.grad, x64.grad, 2e-5, "dx")
    _close(conv.weight.grad, w64.grad, 5e-5, "dw")
    _close(conv.bias.grad, b64.grad, 5e-5, "dbias")


def test_encoder_with_own_convolutions_is_as_accurate_as_the_stock_path(monkeypatch):
    """the whole image encoder (ResNet18 trunk + U-decoder, B = 4, 256 x 256), forward features and the gradient of every
    parameter, three ways on the same weights: (a) own convolutions with fused BatchNorm statistics, (b) the MIOpen +
    separate-BatchNorm composition in fp32, (c) the stock composition in float64.  Two fp32 evaluations of a 20-layer ReLU
    network differ by the rounding of their summation orders (and by ReLU outputs within rounding of zero taking different
    branches), so (a) is not asserted against (b) but against (c): its deviation may not exceed 2 x that of (b) plus a floor."""
    import scp_amd.dino as dino
    from scp_amd import fused_conv
    from scp_amd.flags import Options
    from scp_amd.encoder import Encoder
    dino.ALLOW_RANDOM_INIT = True
    torch.manual_seed(int(os.environ.get("SCP_TEST_SEED", "0")))
    opts = Options("laptop_wild6d", batch_size=2, repeat=2, train=True)
    enc = Encoder(opts).cuda().train()
    enc.backbone.to(memory_format=torch.channels_last)
    enc.featnet.to(memory_format=torch.channels_last)
    enc.random_jitter = torch.nn.Identity()
    img = torch.rand(4, 3, 256, 256, device="cuda")
    state = {k: v.clone() for k, v in enc.state_dict().items()}

    def run(net, x):
        net.zero_grad(set_to_none=True)
        code, feat = net.encode_img(x)
        loss = (feat * torch.linspace(-1, 1, feat.numel(), device="cuda", dtype=feat.dtype).view_as(feat)).sum() + code.square().sum()
        loss.backward()
        return (feat.detach().double(), code.detach().double(),
                {k: p.grad.detach().double() for k, p in net.named_parameters() if p.grad is not None})
    enc.load_state_dict(state)
    feat_a, code_a, grads_a = run(enc, img)
    monkeypatch.setattr(fused_conv, "own_forward_ok", lambda *a, **k: False)          # (b): the fused ops switched off
    enc.load_state_dict(state)
    feat_b, code_b, grads_b = run(enc, img)
    enc.load_state_dict(state)
    enc64 = enc.double()                                                               # (c)
    feat_c, code_c, grads_c = run(enc64, img.double())
    rel = lambda u, v: ((u - v).norm() / v.norm().clamp_min(1e-30)).item()
    print("features: own %.2e stock %.2e (relative L2 vs float64)" % (rel(feat_a, feat_c), rel(feat_b, feat_c)))
    assert rel(feat_a, feat_c) <= 1.5 * rel(feat_b, feat_c) + 2e-6 and rel(code_a, code_c) <= 1.5 * rel(code_b, code_c) + 2e-6
    assert set(grads_a) == set(grads_b) == set(grads_c)
    worst = (0.0, 0.0, "")
    flipped = []
    for k in grads_c:
        ea, eb = rel(grads_a[k], grads_c[k]), rel(grads_b[k], grads_c[k])
        if ea > worst[0]:
            worst = (ea, eb, k)
        # ReLU outputs within rounding of zero take different branches in different evaluations: a single flip in a deep layer
        # moves a small bias gradient by 1e-3 of its norm in either path, hence the additive floor.  At B = 4 layer4 has 256
        # pixels per channel: one flipped unit there shifts the BatchNorm gradients around it by several 1e-3 (all channels a
        # little: the unit feeds every output channel), and with it the gradients of every layer the changed mask back-propagates
        # into.  Whether a seed has such a unit is chance (seeds 1 and 2 have none in either path, seeds 0 and 3 one in the split
        # path); those parameters are reported and held to 2e-2 -- a wrong tap, stride or operand plane is an O(1) error -- while
        # the exact arithmetic of the ops is asserted by the mask-consistent tests above.
        if ea > 2.0 * eb + 1e-3:
            assert ea <= 2e-2, "%s: own %.3e vs stock %.3e (relative L2 vs float64)" % (k, ea, eb)
            flipped.append((k, ea, eb))
    print("parameters beyond the tight bound (behind a flipped ReLU): %d of %d, worst %.2e" % (
        len(flipped), len(grads_c), max([f[1] for f in flipped], default=0.0)))
    print("encoder gradients, worst parameter %s: own %.2e, stock %.2e (relative L2 vs float64)" % (worst[2], worst[0], worst[1]))


def test_weight_planes_kernel_equals_the_torch_composition():
    """scp_conv_weight_planes: both operand-plane sets of a convolution weight in one launch == split_planes of the
    channels_last weight (forward) and of the flipped / transposed weight (input gradient), for contiguous and channels_last
    parameter storage; and the planes sum back to the weight exactly"""
    from scp_amd import fused_conv
    for fmt in (torch.contiguous_format, torch.channels_last):
        conv = nn.Conv2d(64, 96, 3, padding=1, bias=False).cuda().to(memory_format=fmt)
        with torch.no_grad():
            conv.weight.mul_(torch.exp2(torch.randint(-8, 9, conv.weight.shape, device="cuda").float()))
        old, fused_conv.CONV_MODE = fused_conv.CONV_MODE, "split"
        try:
            cache = fused_conv.weight_planes(conv, True)
        finally:
            fused_conv.CONV_MODE = old
        w = conv.weight.detach()
        ref_f = fused_conv.split_planes_tiled(w.permute(0, 2, 3, 1).contiguous())
        ref_d = fused_conv.split_planes_tiled(w.flip(2, 3).permute(1, 2, 3, 0).contiguous())
        assert torch.equal(cache["fwd"], ref_f) and torch.equal(cache["dgrad"], ref_d)
        # tiled layout [rows / 32][K / 16][3][32][16] -> plane-major: the planes sum back to the weight exactly
        from scp_amd.dino import TiledPlanes
        planes = TiledPlanes(96, 9 * 64, "cuda", blob=cache["fwd"]).untile()
        assert torch.equal(planes.double().sum(0).reshape(96, 3, 3, 64), w.permute(0, 2, 3, 1).double())


def test_weight_planes_follow_every_way_a_weight_can_change(conv_mode, monkeypatch):
    """the cached split planes of a convolution (fused_conv.weight_planes) must never serve a stale weight: in-place autograd-visible
    writes (version counter), load_state_dict (module hook), optimizer steps (Optimizers' post-step hook -> invalidate()) and raw
    `.data` writes followed by fused_conv.invalidate(); a raw `.data` write WITHOUT it is what SCP_CHECK_PLANES=1 exists to catch"""
    from scp_amd import fused_conv
    if conv_mode != "split":
        pytest.skip("planes exist in split mode only")
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(2, 64, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)

    def check(what):
        with torch.no_grad():
            got = fused_conv.conv_bias_leaky(x, conv, slope=1.0)
            ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        _close(got, ref, 2e-5, what)

    check("fresh")
    with torch.no_grad():
        conv.weight.mul_(1.5)                                     # in-place, bumps the version counter
    check("in-place write")
    conv.load_state_dict({k: v * 0.5 for k, v in conv.state_dict().items()})
    check("load_state_dict")
    conv.weight.data.add_(0.25)                                   # behind autograd's back
    fused_conv.invalidate()
    check(".data write + invalidate()")
    other = torch.cuda.Stream()
    other.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(other):                                # a consumer on another stream, then a rebuild on the main stream
        check("other stream")
    with torch.no_grad():
        conv.weight.mul_(2.0)
    check("rebuild after a foreign reader")
    torch.cuda.synchronize()
    # the debugging aid: a silent .data write is detected instead of served stale
    monkeypatch.setattr(fused_conv, "CHECK_PLANES", True)
    with torch.no_grad():
        conv.weight.mul_(1.0)                                     # rebuild once with the checksum recorded
    check("checksum armed")
    conv.weight.data.mul_(3.0)
    with pytest.raises(RuntimeError, match="without its split planes"):
        check("stale")


def test_batched_plane_refresh_equals_the_per_layer_launches(conv_mode, monkeypatch):
    """fused_conv.refresh_planes: every stale layer in ONE launch (scp_conv_weight_planes_batch) writes the same bits as one
    scp_conv_weight_planes launch per layer -- contiguous and channels_last weights, 3x3 and 1x1, with and without the input-gradient set"""
    from scp_amd import fused_conv
    if conv_mode != "split":
        pytest.skip("planes exist in split mode only")
    torch.manual_seed(5)
    convs = [nn.Conv2d(64, 64, 3, padding=1), nn.Conv2d(64, 128, 3, stride=2, padding=1), nn.Conv2d(128, 32, 1), nn.Conv2d(256, 512, 3, padding=1),
             nn.Conv2d(64, 48, 3, padding=1), nn.Conv2d(32, 80, 1)]        # the last two: Cout is no multiple of the kernel's 32 x 32 tile
    convs = [c.cuda() for c in convs]
    convs[1].to(memory_format=torch.channels_last)
    convs[3].to(memory_format=torch.channels_last)
    convs[5].to(memory_format=torch.channels_last)
    convs[2].weight.requires_grad_(False)                 # no input-gradient planes wanted for this one
    snaps = {}
    for batch in ("1", "0"):
        monkeypatch.setenv("SCP_PLANES_BATCH", batch)
        for c in convs:
            c.__dict__.pop("_scp_planes", None)
        fused_conv.invalidate()
        fused_conv.refresh_planes(convs)
        torch.cuda.synchronize()
        snaps[batch] = [(c._scp_planes["fwd"].clone(), c._scp_planes["dgrad"].clone() if "dgrad" in c._scp_planes else None) for c in convs]
    for (fa, da), (fb, db) in zip(snaps["1"], snaps["0"]):
        assert torch.equal(fa.view(torch.int16), fb.view(torch.int16))
        assert (da is None) == (db is None) and (da is None or torch.equal(da.view(torch.int16), db.view(torch.int16)))
    assert snaps["1"][2][1] is None and snaps["1"][0][1] is not None
    # and a second refresh after a weight update reuses the cached descriptor table
    monkeypatch.setenv("SCP_PLANES_BATCH", "1")
    with torch.no_grad():
        for c in convs:
            c.weight.mul_(1.25)
    fused_conv.refresh_planes(convs)
    x = torch.randn(2, 64, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = fused_conv.conv_bias_leaky(x, convs[0], slope=1.0)
        ref = F.conv2d(x.double(), convs[0].weight.double(), convs[0].bias.double(), padding=1)
    _close(got, ref, 2e-5, "after batched refresh")
