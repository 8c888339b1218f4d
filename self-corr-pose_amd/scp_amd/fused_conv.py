"""scp_amd/fused_conv.py -- the image encoder's convolutions on the build's own implicit-GEMM kernels (csrc/conv_igemm.hip,
csrc/conv_wgrad.hip), fused with what follows them.

`conv_bn_act(x, conv, bn, skip=None, relu=False)` == `relu(bn(conv(x)) + skip)`: torchvision's BasicBlock as used by
model/module/network/image_encoder.py:119-139.  ONE autograd op:
  forward : convolution whose epilogue leaves the per-tile column sums and whose last workgroup finalises the batch statistics
            (no statistics pass over the activation), then one apply pass (scale, shift [, + skip] [, ReLU]);
  backward: the BatchNorm(+ReLU) backward of csrc/batchnorm.hip, then the input gradient (the forward kernel on flipped /
            transposed weights) and the weight gradient (halo-block kernel, deterministic).
`conv_bias_leaky(x, conv, slope, stride)` == `leaky_relu(conv(x) + bias)`: the decoder's conv unit
(net_blocks.py:336-359 with_bn=False); bias and activation live in the convolution's epilogue.

What the own kernels do not cover goes to MIOpen through ATen, layer by layer and direction by direction, never silently for a
whole network.  The 7x7 stem (Cin = 3) has its own kernels (csrc/conv_stem.hip, `stem_conv_bn_act`).  The stride-2 layers' backward runs on the split main loop (round 4: input gradient by
parity class, scp_conv_nhwc_dgrad_stride2; weight gradient of the 3x3 and of the 1x1 projections by the halo-block kernel with a
strided block); with SCP_CONV_GEMM=fp32 / SCP_CONV_WGRAD=fp32 those directions fall back to `aten.convolution_backward` / one
gather + library GEMM.  The 1x1 projections' input gradient is an own 1x1 product scattered to the even pixels.  CPU tensors, eval-mode BatchNorm, SyncBatchNorm and non-fp32 activations (configs[4] bf16
autocast) take the stock composition, which is also what the tests compare with."""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from . import capi


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _pow2(v):
    return v > 0 and (v & (v - 1)) == 0


def own_forward_ok(x, weight, stride):
    """shapes scp_conv_nhwc_forward covers: 3x3 / 1x1, stride 1 / 2, Cin a power of two >= 32"""
    k = weight.shape[2]
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and k == weight.shape[3]
            and k in (1, 3) and stride in (1, 2) and weight.shape[1] >= 32 and _pow2(weight.shape[1]))


def _own_dgrad_ok(weight, stride):
    """stride 1; the 1x1 stride-2 projections of the ResNet trunk (their input gradient is a 1x1 convolution of dy scattered to
    the even pixels); the 3x3 stride-2 layers on the split main loop (scp_conv_nhwc_dgrad_stride2: by input-pixel parity class)"""
    cout, k = weight.shape[0], weight.shape[2]
    return (stride == 1 or (stride == 2 and (k == 1 or CONV_MODE == "split"))) and cout >= 32 and _pow2(cout)


def _own_wgrad_ok(x_shape, weight, stride):
    """3x3 stride 1; on the split core also the stride-2 layers of the trunk (3x3 conv1 and the 1x1 projection of layer2..4) and 1x1
    stride 1 (the decoder's feature projection)"""
    n, cin, h, w = x_shape
    k = weight.shape[2]
    if not ((stride == 1 and k == 3) or (WGRAD_MODE == "split" and k == 1 and stride == 1)
            or (stride == 2 and k in (1, 3) and WGRAD_MODE == "split" and h % 2 == 0 and w % 2 == 0)):
        return False
    ho, wo = h // stride, w // stride
    return (ho >= 8 and wo >= 8 and _pow2(ho) and _pow2(wo) and cin % 64 == 0 and weight.shape[0] % 64 == 0
            and (n * ho * wo) % 32 == 0)


def _conv_out_shape(x, weight, stride):
    n, _, h, w = x.shape
    k = weight.shape[2]
    return n, weight.shape[0], (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1


# "split": the convolutions' products (forward, input gradient, weight gradient) run on the bf16 matrix cores with exactly split
# operands (csrc/gemm_core_split.h: every fp32 value = three bf16 terms, six partial products, fp32 accumulation -- fp32-accurate,
# ~1.4x the fp32 cores' rate at the part's power limit).  "fp32": v_mfma_f32_32x32x2_f32.  SCP_CONV_GEMM=fp32 in the environment
# selects the latter.
import os
CONV_MODE = os.environ.get("SCP_CONV_GEMM", "split")
# the weight gradient's own switch (csrc/conv_wgrad.hip: both operands are activations, split in registers): "split" | "fp32"
WGRAD_MODE = os.environ.get("SCP_CONV_WGRAD", CONV_MODE)


def split_planes(t):
    """planes [3, *t.shape] bf16 with t = h + m + l exactly (scp_split_bf16x3); t contiguous fp32"""
    planes = torch.empty((3,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
    capi.check(capi.lib().scp_split_bf16x3(_ptr(t), _ptr(planes), t.numel(), capi.current_stream()), "split_bf16x3")
    return planes


def tiled_planes_numel(rows, k):
    return 3 * ((rows + 31) // 32 * 32) * k


def split_planes_tiled(t):
    """the W operand of the split convolution kernels from a weight already in [rows, ...K] order (e.g. [Cout, k, k, Cin]): flat
    bf16 storage in the TILED plane layout (csrc/gemm_core_split.h; scp_split_bf16x3_tiled)"""
    rows = t.shape[0]
    k = t.numel() // rows
    planes = torch.empty(tiled_planes_numel(rows, k), dtype=torch.bfloat16, device=t.device)
    capi.check(capi.lib().scp_split_bf16x3_tiled(_ptr(t.contiguous()), _ptr(planes), rows, k, capi.current_stream()), "split_bf16x3_tiled")
    return planes


def weight_planes(conv, with_dgrad):
    """The split operands of a convolution's weight for this weight version, kept on the module: "fwd" = planes of
    [Cout, k, k, Cin], "dgrad" = planes of the flipped / transposed [Cin, k, k, Cout] the input gradient multiplies with, both in the
    tiled plane layout (flat bf16 storage).
    Both encoder passes of a step (and their backward passes) share them.  The buffers are PERSISTENT (their addresses are baked into
    the HIP graphs of scp_amd.graphed) and rebuilt IN PLACE when the weight has changed: lazily here, keyed by (storage, tensor
    version, WEIGHT_EPOCH), or eagerly by refresh_planes(), which Trainer.step calls once per step on the main stream before any side
    stream starts (updates made through `.data` -- optimizer-free loading, broadcasts -- do not bump the tensor version: callers of
    such updates bump WEIGHT_EPOCH).  Under graph capture a stale cache is an error, not a launch."""
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device), WEIGHT_EPOCH[0])
    cache = conv.__dict__.get("_scp_planes")
    if cache is None:
        cache = conv.__dict__["_scp_planes"] = {"key": None, "readers": {}}
        # whoever loads a state dict into the module writes the weight without touching the key above
        conv.register_load_state_dict_post_hook(lambda module, incompatible: invalidate())
    cur = torch.cuda.current_stream(w.device)
    stale = cache["key"] != key
    if stale or (with_dgrad and "dgrad" not in cache):
        if capi.CAPTURING:
            raise RuntimeError("scp_amd.fused_conv: weight planes are stale under graph capture -- call refresh_planes() first")
        cout, cin, k, _ = w.shape
        if "fwd" not in cache:
            cache["fwd"] = torch.zeros(tiled_planes_numel(cout, k * k * cin), dtype=torch.bfloat16, device=w.device)
        build_dgrad = with_dgrad or "dgrad" in cache
        if build_dgrad and "dgrad" not in cache:
            cache["dgrad"] = torch.zeros(tiled_planes_numel(cin, k * k * cout), dtype=torch.bfloat16, device=w.device)
        # the buffers are rewritten in place: streams that read the previous contents (side-stream encoder pass, look-ahead) finish first
        for other in cache["readers"].values():
            cur.wait_stream(other)
        cache["readers"] = {}
        wd = w.detach()
        # one launch fills both plane sets (a "fwd" set that is current is rewritten with the values it already holds)
        capi.check(capi.lib().scp_conv_weight_planes(ctypes.c_void_p(wd.data_ptr()), wd.stride(0), wd.stride(1), wd.stride(2), wd.stride(3),
                                                     cout, cin, k, _ptr(cache["fwd"]), _ptr(cache["dgrad"] if build_dgrad else None),
                                                     capi.current_stream()), "conv_weight_planes")
        cache["key"] = key
        cache["built_on"] = cur
        cache["built"] = torch.cuda.Event()
        cache["built"].record(cur)
        if CHECK_PLANES:
            cache["checksum"] = wd.double().sum()
    elif cache.get("built_on") is not None and cur != cache["built_on"] and not capi.CAPTURING:
        # a consumer on another stream than the one that built them: ordered behind the build, remembered for the next rebuild
        if cur.cuda_stream not in cache["readers"]:
            cur.wait_event(cache["built"])
            cache["readers"][cur.cuda_stream] = cur
    if CHECK_PLANES and not capi.CAPTURING and "checksum" in cache and not torch.equal(cache["checksum"], w.detach().double().sum()):
        raise RuntimeError("scp_amd.fused_conv: the weight of %r changed without its split planes being rebuilt (a write through .data? "
                           "call fused_conv.invalidate() after such writes)" % (conv,))
    return cache


# SCP_CHECK_PLANES=1: every use of cached planes compares a checksum of the weight with the one taken when they were built (one host
# sync per convolution call: a debugging aid for code that updates weights behind autograd's back)
CHECK_PLANES = os.environ.get("SCP_CHECK_PLANES", "0") == "1"


def invalidate():
    """Every cached plane set is rebuilt at its next use.  Call after writing convolution weights in a way that neither bumps the
    tensor's version counter nor goes through load_state_dict (both are noticed automatically): `.data` writes -- EMA updates,
    vector_to_parameters, hand-written broadcasts, optimizers that step through `.data` (scp_amd.optimizers registers this as a
    post-step hook for exactly that case)."""
    WEIGHT_EPOCH[0] += 1


# bumped by whoever changes weights without going through an optimizer step on the tensor itself (load_state_dict / broadcast via .data)
WEIGHT_EPOCH = [0]


def refresh_planes(convs):
    """rebuild the split planes of every convolution in `convs` whose weight changed since they were built (Trainer.step: once per
    step, main stream, before the forward).  After an optimizer step that is every layer: they go out as ONE launch
    (scp_conv_weight_planes_batch; the ~27 single launches sat on the critical path between the optimizer and the forward)."""
    if CONV_MODE != "split":
        return
    todo = []
    for conv in convs:
        w = conv.weight
        # only the layers the own kernels run (own_forward_ok's shape rules: 3x3 / 1x1, Cin a power of two >= 32)
        if (w.is_cuda and w.dtype == torch.float32 and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3) and w.shape[1] >= 32 and _pow2(w.shape[1])
                and w.shape[0] % 2 == 0):          # even Cout: the batched kernel stores pairs of K positions (include/scp_hip.h)
            cache = conv.__dict__.get("_scp_planes", {})
            with_dgrad = (w.requires_grad and _own_dgrad_ok(w, conv.stride[0])) or "dgrad" in cache
            key = (w.data_ptr(), w._version, str(w.device), WEIGHT_EPOCH[0])
            if cache.get("key") != key or (with_dgrad and "dgrad" not in cache):
                todo.append((conv, with_dgrad, key))
    if len(todo) < 2 or capi.CAPTURING or CHECK_PLANES or os.environ.get("SCP_PLANES_BATCH", "1") != "1":
        for conv, with_dgrad, _ in todo:
            weight_planes(conv, with_dgrad=with_dgrad)
        return
    dev = todo[0][0].weight.device
    cur = torch.cuda.current_stream(dev)
    descs = []
    for conv, with_dgrad, key in todo:
        w = conv.weight
        cache = conv.__dict__.get("_scp_planes")
        if cache is None:
            cache = conv.__dict__["_scp_planes"] = {"key": None, "readers": {}}
            conv.register_load_state_dict_post_hook(lambda module, incompatible: invalidate())
        cout, cin, k, _ = w.shape
        if "fwd" not in cache:
            cache["fwd"] = torch.zeros(tiled_planes_numel(cout, k * k * cin), dtype=torch.bfloat16, device=w.device)
        if with_dgrad and "dgrad" not in cache:
            cache["dgrad"] = torch.zeros(tiled_planes_numel(cin, k * k * cout), dtype=torch.bfloat16, device=w.device)
        for other in cache["readers"].values():        # streams that read the previous contents finish first (see weight_planes)
            cur.wait_stream(other)
        cache["readers"] = {}
        descs.append((w.data_ptr(), cache["fwd"].data_ptr(), cache["dgrad"].data_ptr() if "dgrad" in cache else 0,
                      w.stride(0), w.stride(1), w.stride(2), w.stride(3), cout, cin, k))
    sig = tuple(descs)
    table = _BATCH_TABLES.get(sig)
    if table is None:
        arr = (capi.ConvPlanesDesc * len(descs))()
        block0 = 0
        for e, d in zip(arr, descs):
            e.w, e.planes_fwd, e.planes_dgrad, e.s_co, e.s_ci, e.s_ky, e.s_kx, e.Cout, e.Cin, e.ksize = d
            e.block0 = block0
            block0 += -(-d[7] // capi.CONV_PLANES_TILE) * -(-d[8] // capi.CONV_PLANES_TILE)      # one workgroup per (Cout, Cin) tile
        host = torch.empty(ctypes.sizeof(arr), dtype=torch.uint8)
        ctypes.memmove(host.data_ptr(), ctypes.addressof(arr), host.numel())
        _BATCH_TABLES.clear()                          # pointers of a previous model are of no use to anyone
        table = _BATCH_TABLES[sig] = (host.to(dev), len(descs), block0)
    capi.check(capi.lib().scp_conv_weight_planes_batch(ctypes.c_void_p(table[0].data_ptr()), table[1], table[2], capi.current_stream()),
               "conv_weight_planes_batch")
    built = torch.cuda.Event()
    built.record(cur)
    for conv, _, key in todo:
        cache = conv.__dict__["_scp_planes"]
        cache["key"], cache["built_on"], cache["built"] = key, cur, built


_BATCH_TABLES = {}


def _planes_arg(conv, x, stride):
    """what the fused ops get as their `planes` argument: the module's cache in split mode, None for the fp32 cores"""
    if CONV_MODE != "split":
        return None
    need_dx = torch.is_grad_enabled() and x.requires_grad and _own_dgrad_ok(conv.weight, stride)
    return weight_planes(conv, need_dx)


def _splitk(n, h, w, cin, cout, k, stride, split, device):
    """(buffer, bytes) for the split-K partial tiles of this layer (include/scp_hip.h: scp_conv_nhwc_splitk_workspace); (None, 0)
    when the layer does not split"""
    nbytes = capi.lib().scp_conv_nhwc_splitk_workspace(n, h, w, cin, cout, k, stride, int(split))
    if not nbytes:
        return None, 0
    return torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes


def _conv_forward(x, weight, bias, stride, leaky, slope, planes=None):
    """raw own forward: x, weight channels_last; returns a channels_last tensor"""
    L = capi.lib()
    n, cin, h, w = x.shape
    cout, k = weight.shape[0], weight.shape[2]
    y = torch.empty(_conv_out_shape(x, weight, stride), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    sk, sk_bytes = _splitk(n, h, w, cin, cout, k, stride, planes is not None, x.device)
    capi.check(L.scp_conv_nhwc_forward(_ptr(x), _ptr(None if planes else weight), _ptr(planes["fwd"] if planes else None), _ptr(bias),
                                       _ptr(y), _ptr(None), n, h, w, cin, cout, k, stride, int(leaky), float(slope), _ptr(sk), sk_bytes,
                                       capi.current_stream()), "conv_nhwc_forward")
    return y


def _conv_backward(x, weight, g, stride, need_dx, need_dw, planes=None):
    """(dx, dw) of y = conv(x, weight) for the output gradient g; all channels_last.  Own kernels where they apply."""
    L = capi.lib()
    n, cin, h, w = x.shape
    cout, k = weight.shape[0], weight.shape[2]
    dx = dw = None
    own_dx = need_dx and _own_dgrad_ok(weight, stride)
    own_dw = need_dw and _own_wgrad_ok(x.shape, weight, stride)
    if own_dx and stride == 2 and k == 3:
        own_dx = planes is not None and "dgrad" in planes and h % 2 == 0 and w % 2 == 0
        if own_dx:
            dx = torch.empty(x.shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            capi.check(L.scp_conv_nhwc_dgrad_stride2(_ptr(g), _ptr(planes["dgrad"]), _ptr(dx), n, h // 2, w // 2, cout, cin,
                                                     capi.current_stream()), "conv_nhwc_dgrad_stride2")
    elif own_dx:
        # the forward kernel on dy with the weights as [Cin, k, k, Cout], taps flipped
        wt3 = planes.get("dgrad") if planes else None
        wt = None if wt3 is not None else weight.flip(2, 3).permute(1, 2, 3, 0).contiguous()
        ho, wo = g.shape[2], g.shape[3]
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if stride == 1:
            sk, sk_bytes = _splitk(n, ho, wo, cout, cin, k, 1, wt3 is not None, x.device)
            capi.check(L.scp_conv_nhwc_forward(_ptr(g), _ptr(wt), _ptr(wt3), _ptr(None), _ptr(dx), _ptr(None), n, ho, wo, cout, cin, k, 1,
                                               0, 0.0, _ptr(sk), sk_bytes, capi.current_stream()), "conv_nhwc_forward (input gradient)")
        elif h == 2 * ho and w == 2 * wo:
            # stride 2 (1x1 only): the same product at the output resolution, written to the even pixels by the epilogue, zeros between
            capi.check(L.scp_conv1x1_nhwc_dgrad_stride2(_ptr(g), _ptr(wt), _ptr(wt3), _ptr(dx), n, ho, wo, cout, cin, capi.current_stream()),
                       "conv1x1_nhwc_dgrad_stride2")
        else:                                            # odd maps: the product, then a strided copy into a zero gradient
            dense = torch.empty((n, cin, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            capi.check(L.scp_conv_nhwc_forward(_ptr(g), _ptr(wt), _ptr(wt3), _ptr(None), _ptr(dense), _ptr(None), n, ho, wo, cout, cin, k, 1,
                                               0, 0.0, _ptr(None), 0, capi.current_stream()), "conv_nhwc_forward (input gradient)")
            dx.zero_()
            dx[:, :, ::2, ::2] = dense
    if own_dw:
        ws_bytes = L.scp_conv_nhwc_weight_grad_workspace(n, h, w, cin, cout, k, stride)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        capi.check(L.scp_conv_nhwc_weight_grad(_ptr(x), _ptr(g), _ptr(dw), _ptr(None), _ptr(ws), ws_bytes, n, h, w, cin, cout, k, stride,
                                               int(WGRAD_MODE == "split"), capi.current_stream()), "conv_nhwc_weight_grad")
    if need_dw and not own_dw and k == 1 and stride == 2:
        # weight gradient of a 1x1 stride-2 projection = dy^T (Cout x pixels) @ x at the even pixels (pixels x Cin): one gather + one
        # library GEMM instead of MIOpen's 50-60 us kernels for 0.5 GFLOP
        xs = x.permute(0, 2, 3, 1)[:, ::2, ::2, :].reshape(-1, cin)
        g2 = g.permute(0, 2, 3, 1).reshape(-1, cout)
        dw = torch.mm(g2.t(), xs).reshape(cout, cin, 1, 1)
        own_dw = True
    miss_dx, miss_dw = need_dx and not own_dx, need_dw and not own_dw
    if miss_dx or miss_dw:
        p = weight.shape[2] // 2
        rdx, rdw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, [stride, stride], [p, p], [1, 1], False, [0, 0], 1,
                                                          [miss_dx, miss_dw, False])
        if miss_dx:
            dx = rdx
        if miss_dw:
            dw = rdw
    return dx, dw


class _ConvBNAct(Function):
    @staticmethod
    def forward(ctx, x, weight, skip, gamma, beta, bn, relu, stride, planes):
        L = capi.lib()
        x = _nhwc(x)
        weight = weight if planes else _nhwc(weight)      # split mode: the kernels read the planes, `weight` only gives shapes
        n, cin, h, w = x.shape
        cout, k = weight.shape[0], weight.shape[2]
        conv = torch.empty(_conv_out_shape(x, weight, stride), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        rows = conv.shape[0] * conv.shape[2] * conv.shape[3]
        stats = torch.empty(4, cout, dtype=torch.float32, device=x.device)
        tiles = ctypes.c_int()
        L.scp_conv_nhwc_partial_rows(n, h, w, cin, cout, k, stride, int(planes is not None), ctypes.byref(tiles), None)
        ws = torch.empty(2 * tiles.value * cout, dtype=torch.float32, device=x.device)
        momentum = 0.1 if bn.momentum is None else bn.momentum
        sk, sk_bytes = _splitk(n, h, w, cin, cout, k, stride, planes is not None, x.device)
        capi.check(L.scp_conv_nhwc_forward_bn(
            _ptr(x), _ptr(None if planes else weight), _ptr(planes["fwd"] if planes else None), _ptr(conv), n, h, w, cin, cout, k,
            stride, _ptr(gamma), _ptr(beta), _ptr(bn.running_mean),
            _ptr(bn.running_var), _ptr(bn.num_batches_tracked if bn.track_running_stats else None), float(momentum), float(bn.eps),
            _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(ws), ws.numel() * 4, capi.ticket(x.device),
            _ptr(sk), sk_bytes, capi.current_stream()), "conv_nhwc_forward_bn")
        if skip is not None:
            skip = _nhwc(skip)
        y = torch.empty_like(conv)
        capi.check(L.scp_batchnorm_apply(_ptr(conv), _ptr(skip), _ptr(stats[2]), _ptr(stats[3]), rows, cout, int(relu), _ptr(y),
                                         capi.current_stream()), "batchnorm_apply")
        residual_relu = relu and skip is not None
        ctx.save_for_backward(x, weight, conv, y if residual_relu else None, stats)
        ctx.cfg = (rows, cout, bool(relu), skip is not None, gamma is not None, beta is not None, stride)
        ctx.planes = planes
        return y

    @staticmethod
    def backward(ctx, dy):
        L = capi.lib()
        x, weight, conv, y, stats = ctx.saved_tensors
        rows, c, relu, has_skip, has_w, has_b, stride = ctx.cfg
        dy = _nhwc(dy)
        dconv = torch.empty_like(conv)
        dskip = torch.empty_like(conv) if (relu and has_skip) else None
        want_g = has_w and ctx.needs_input_grad[3]
        want_b = has_b and ctx.needs_input_grad[4]
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device) if want_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device) if want_b else None
        ws_bytes = L.scp_batchnorm_workspace(rows, c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        capi.check(L.scp_batchnorm_act_backward(
            _ptr(dy), _ptr(conv), _ptr(y), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), rows, c, int(relu),
            int(has_skip), 1, _ptr(dconv), _ptr(dskip), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws_bytes, capi.ticket(x.device),
            capi.current_stream()), "batchnorm_act_backward")
        if has_skip and dskip is None:
            dskip = dy
        dx, dw = _conv_backward(x, weight, dconv, stride, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.planes)
        return dx, dw, (dskip if has_skip else None), dgamma, dbeta, None, None, None, None


class _StemConvBNAct(Function):
    """relu(bn(conv7x7s2(x))) of the ResNet stem (image_encoder.py:122-124) on csrc/conv_stem.hip: the convolution leaves the batch
    statistics (one launch), one apply pass; backward = the BatchNorm(+ReLU) backward and the stem's weight gradient.  The image
    carries no gradient."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, relu):
        L = capi.lib()
        x = x.contiguous()                                  # NCHW: the kernel reads image rows
        n, _, h, w = x.shape
        cout = weight.shape[0]
        conv = torch.empty((n, cout, h // 2, w // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        rows = n * (h // 2) * (w // 2)
        stats = torch.empty(4, cout, dtype=torch.float32, device=x.device)
        ws = torch.empty(2 * L.scp_stem_conv_tiles(n, h, w) * cout, dtype=torch.float32, device=x.device)
        momentum = 0.1 if bn.momentum is None else bn.momentum
        wd = weight.detach()
        capi.check(L.scp_stem_conv_forward_bn(
            _ptr(x), _ptr(wd), wd.stride(0), wd.stride(1), wd.stride(2), wd.stride(3), _ptr(conv), n, h, w, _ptr(gamma), _ptr(beta),
            _ptr(bn.running_mean), _ptr(bn.running_var), _ptr(bn.num_batches_tracked if bn.track_running_stats else None),
            float(momentum), float(bn.eps), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(ws), ws.numel() * 4,
            capi.ticket(x.device), capi.current_stream()), "stem_conv_forward_bn")
        y = torch.empty_like(conv)
        capi.check(L.scp_batchnorm_apply(_ptr(conv), _ptr(None), _ptr(stats[2]), _ptr(stats[3]), rows, cout, int(relu), _ptr(y),
                                         capi.current_stream()), "batchnorm_apply")
        ctx.save_for_backward(x, weight, conv, stats)
        ctx.cfg = (rows, cout, bool(relu), gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = capi.lib()
        x, weight, conv, stats = ctx.saved_tensors
        rows, c, relu, has_w, has_b = ctx.cfg
        dy = _nhwc(dy)
        dconv = torch.empty_like(conv)
        want_g = has_w and ctx.needs_input_grad[2]
        want_b = has_b and ctx.needs_input_grad[3]
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device) if want_g else None
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device) if want_b else None
        ws_bytes = L.scp_batchnorm_workspace(rows, c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        capi.check(L.scp_batchnorm_act_backward(
            _ptr(dy), _ptr(conv), _ptr(None), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), rows, c, int(relu),
            0, 1, _ptr(dconv), _ptr(None), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws_bytes, capi.ticket(x.device),
            capi.current_stream()), "batchnorm_act_backward")
        dw = None
        if ctx.needs_input_grad[1]:
            n, _, h, w = x.shape
            dw = torch.empty_like(weight)
            wg_bytes = L.scp_stem_conv_weight_grad_workspace(n, h, w)
            wg = torch.empty(wg_bytes // 4, dtype=torch.float32, device=x.device)
            capi.check(L.scp_stem_conv_weight_grad(_ptr(x), _ptr(dconv), _ptr(dw), dw.stride(0), dw.stride(1), dw.stride(2), dw.stride(3),
                                                   _ptr(wg), wg_bytes, n, h, w, capi.current_stream()), "stem_conv_weight_grad")
        return None, dw, dgamma, dbeta, None, None


def stem_takes_own_kernels(x, conv, bn, autocast=None, meta=None):
    """whether stem_conv_bn_act(x, conv, bn) will run csrc/conv_stem.hip -- which reads the image as NCHW rows; the encoder asks before
    choosing the layout of the normalised image (scp_amd/encoder.py), so that no layout copy is made only to be undone.
    `meta` = (shape, dtype, is_cuda, requires_grad) stands in for a tensor that does not exist yet (x is then ignored): the question is
    about metadata only, nothing is converted to answer it."""
    w = conv.weight
    shape, dtype, is_cuda, requires_grad = meta if meta is not None else (tuple(x.shape), x.dtype, x.is_cuda, x.requires_grad)
    autocast = torch.is_autocast_enabled() if autocast is None else autocast
    return (is_cuda and dtype == torch.float32 and w.dtype == torch.float32 and len(shape) == 4 and tuple(w.shape) == (64, 3, 7, 7)
            and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and shape[1] == 3 and shape[2] % 2 == 0 and shape[3] % 4 == 0 and shape[2] >= 8 and shape[3] >= 8
            and not requires_grad and type(bn) is nn.BatchNorm2d and (bn.training or bn.running_mean is None)
            and not autocast and os.environ.get("SCP_STEM", "own") == "own")


def stem_conv_bn_act(x, conv, bn, relu=True):
    """relu(bn(conv(x))) for the 7x7 / stride-2 / pad-3 stem of the ResNet trunk; anything else (CPU, eval-mode BatchNorm, autocast,
    an image that needs a gradient, odd sizes) takes the stock composition -- on channels_last input, which is what the layers behind
    it run in (a caller that guessed NCHW for the own kernel and was wrong pays one layout copy here, not one per layer)"""
    from .fused_bn import bn_act
    if stem_takes_own_kernels(x, conv, bn):
        return _StemConvBNAct.apply(x, conv.weight, bn.weight, bn.bias, bn, relu)
    if x.is_cuda and x.dim() == 4 and conv.weight.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return bn_act(conv(x), bn, relu=relu)


class _ConvBiasLeaky(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, slope, stride, planes):
        x = _nhwc(x)
        weight = weight if planes else _nhwc(weight)
        y = _conv_forward(x, weight, bias, stride, True, slope, planes)
        ctx.save_for_backward(x, weight, y)
        ctx.cfg = (float(slope), stride)
        ctx.planes = planes
        return y

    @staticmethod
    def backward(ctx, dy):
        L = capi.lib()
        x, weight, y = ctx.saved_tensors
        slope, stride = ctx.cfg
        dy = _nhwc(dy)
        n, c, h, w = y.shape
        rows = n * h * w
        g = torch.empty_like(y)
        dbias = torch.empty(c, dtype=torch.float32, device=y.device) if ctx.needs_input_grad[2] else None
        ws_bytes = L.scp_batchnorm_workspace(rows, c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=y.device)
        # gradient of the pre-activation (mask from the sign of the output: slope > 0 keeps it) + bias gradient, one pass
        capi.check(L.scp_bias_leaky_relu_backward(_ptr(dy), _ptr(y), slope, rows, c, _ptr(g), _ptr(dbias), _ptr(ws), ws_bytes,
                                                  capi.ticket(y.device), capi.current_stream()), "bias_leaky_relu_backward")
        dx, dw = _conv_backward(x, weight, g, stride, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.planes)
        return dx, dw, dbias, None, None, None


def _fused_ok(x, conv, stride):
    return (own_forward_ok(x, conv.weight, stride) and conv.groups == 1 and conv.dilation == (1, 1)
            and conv.padding == (conv.kernel_size[0] // 2, conv.kernel_size[0] // 2) and x.shape[1] == conv.weight.shape[1])


def conv_bn_act(x, conv, bn, skip=None, relu=False):
    """relu(bn(conv(x)) + skip) with nn.Conv2d (no bias) / nn.BatchNorm2d semantics"""
    from .fused_bn import bn_act
    stride = conv.stride[0]
    c = conv.weight.shape[0]
    if (conv.bias is None and _fused_ok(x, conv, stride) and type(bn) is nn.BatchNorm2d and (bn.training or bn.running_mean is None)
            and 16 <= c <= 1024 and _pow2(c) and (skip is None or skip.dtype == torch.float32) and not torch.is_autocast_enabled()):
        return _ConvBNAct.apply(x, conv.weight, skip, bn.weight, bn.bias, bn, relu, stride, _planes_arg(conv, x, stride))
    return bn_act(conv(x), bn, skip=skip, relu=relu)


def conv_bias_leaky(x, conv, slope=0.1, stride=None):
    """leaky_relu(conv(x), slope) for an nn.Conv2d with bias (stride override: the same weights as a strided convolution)"""
    from . import fused_bn
    stride = conv.stride[0] if stride is None else stride
    c = conv.weight.shape[0]
    if (conv.bias is not None and _fused_ok(x, conv, stride) and 16 <= c <= 1024 and _pow2(c) and slope > 0
            and not torch.is_autocast_enabled()):
        return _ConvBiasLeaky.apply(x, conv.weight, conv.bias, slope, stride, _planes_arg(conv, x, stride))
    return fused_bn.conv_bias_leaky(x, conv, slope, stride)
