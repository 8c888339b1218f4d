"""scp_amd/fused_bn.py -- BatchNorm2d (+ residual) (+ ReLU) of the ResNet18 trunk as one autograd op.

`bn_act(x, bn, skip=None, relu=False)` == `relu(bn(x) + skip)` with nn.BatchNorm2d semantics
(model/module/network/image_encoder.py:119-139 runs torchvision's BasicBlock).  On the GPU with NHWC fp32
activations this is csrc/batchnorm.hip (3-4 passes over the activation forward, 5-7 backward, instead of
5-8 and 8 through MIOpen + ReLU + add); there is no fallback for that case.  Everything else -- CPU
tensors (golden runs, CPU baseline), SyncBatchNorm, other dtypes/layouts -- takes the stock torch
composition, which is also what the tests compare the kernels with."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import capi


def _composition(x, bn, skip, relu):
    y = bn(x)
    if skip is not None:
        y = y + skip
    return F.relu(y) if relu else y


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, skip, weight, bias, bn, relu):
        L = capi.lib()
        x = _nhwc(x)
        n, c, h, w = x.shape
        rows = n * h * w
        if skip is not None:
            skip = _nhwc(skip)
        y = torch.empty_like(x)                                   # keeps channels_last
        stats = torch.empty(4, c, dtype=torch.float32, device=x.device)
        ws_bytes = L.scp_batchnorm_workspace(rows, c)
        if ws_bytes == 0:
            raise RuntimeError("scp_amd: batchnorm kernel supports power-of-two channel counts in [16,1024], got %d" % c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        training = bn.training or bn.running_mean is None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        fwd = L.scp_batchnorm_act_forward if x.dtype == torch.float32 else L.scp_batchnorm_act_forward_bf16
        capi.check(fwd(
            _ptr(x), _ptr(skip), _ptr(weight), _ptr(bias), _ptr(bn.running_mean), _ptr(bn.running_var),
            _ptr(bn.num_batches_tracked if bn.track_running_stats else None), float(momentum), float(bn.eps), rows, c,
            int(relu), int(training), _ptr(y), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]),
            _ptr(ws), ws_bytes, capi.ticket(x.device), capi.current_stream()), "batchnorm_act_forward")
        residual_relu = relu and skip is not None
        ctx.save_for_backward(x, y if residual_relu else None, stats)
        ctx.cfg = (rows, c, bool(relu), skip is not None, bool(training), weight is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = capi.lib()
        x, y, stats = ctx.saved_tensors
        rows, c, relu, has_skip, training, has_w, has_b = ctx.cfg
        dy = _nhwc(dy.to(x.dtype))
        dx = torch.empty_like(x)
        dskip = torch.empty_like(x) if (relu and has_skip) else None
        want_w = has_w and ctx.needs_input_grad[2]
        want_b = has_b and ctx.needs_input_grad[3]
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device) if want_w else None
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device) if want_b else None
        ws_bytes = L.scp_batchnorm_workspace(rows, c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
        bwd = L.scp_batchnorm_act_backward if x.dtype == torch.float32 else L.scp_batchnorm_act_backward_bf16
        capi.check(bwd(
            _ptr(dy), _ptr(x), _ptr(y), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), rows, c,
            int(relu), int(has_skip), int(training), _ptr(dx), _ptr(dskip), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
            ws_bytes, capi.ticket(x.device), capi.current_stream()), "batchnorm_act_backward")
        if has_skip and dskip is None:
            dskip = dy
        return dx, (dskip if has_skip else None), dgamma, dbeta, None, None


def _fused_ok(x, bn, skip):
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 4 and type(bn) is nn.BatchNorm2d
            and x.is_contiguous(memory_format=torch.channels_last)
            and (skip is None or (skip.shape == x.shape and skip.dtype == x.dtype)))


def bn_act(x, bn, skip=None, relu=False):
    if not _fused_ok(x, bn, skip):
        return _composition(x, bn, skip, relu)
    return _BatchNormAct.apply(x, skip, bn.weight, bn.bias, bn, relu)


class _BiasLeakyReLU(Function):
    """y = leaky_relu(y0 + bias, slope) in place on the (bias-free) convolution output y0, NHWC (csrc/batchnorm.hip
    scp_bias_leaky_relu_*): one pass forward, one pass backward (gradient of the pre-activation + bias gradient)"""

    @staticmethod
    def forward(ctx, y0, bias, slope):
        L = capi.lib()
        n, c, h, w = y0.shape
        rows = n * h * w
        fwd = L.scp_bias_leaky_relu_forward if y0.dtype == torch.float32 else L.scp_bias_leaky_relu_forward_bf16
        capi.check(fwd(_ptr(y0), _ptr(bias), float(slope), rows, c, capi.current_stream()), "bias_leaky_relu_forward")
        ctx.mark_dirty(y0)
        ctx.save_for_backward(y0)
        ctx.cfg = (rows, c, float(slope))
        return y0

    @staticmethod
    def backward(ctx, dy):
        L = capi.lib()
        (y,) = ctx.saved_tensors
        rows, c, slope = ctx.cfg
        dy = _nhwc(dy.to(y.dtype))
        g = torch.empty_like(y)
        want_b = ctx.needs_input_grad[1]
        dbias = torch.empty(c, dtype=torch.float32, device=y.device) if want_b else None
        ws_bytes = L.scp_batchnorm_workspace(rows, c)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=y.device)
        bwd = L.scp_bias_leaky_relu_backward if y.dtype == torch.float32 else L.scp_bias_leaky_relu_backward_bf16
        capi.check(bwd(_ptr(dy), _ptr(y), slope, rows, c, _ptr(g), _ptr(dbias), _ptr(ws), ws_bytes, capi.ticket(y.device),
                       capi.current_stream()), "bias_leaky_relu_backward")
        return g, dbias, None


def bias_leaky_ok(y0, bias):
    c = y0.shape[1] if y0.dim() == 4 else 0
    return (y0.is_cuda and y0.dtype in (torch.float32, torch.bfloat16) and y0.dim() == 4 and bias is not None
            and bias.dtype == torch.float32 and 16 <= c <= 1024 and (c & (c - 1)) == 0
            and y0.is_contiguous(memory_format=torch.channels_last))


def conv_bias_leaky(x, conv, slope=0.1, stride=None):
    """leaky_relu(conv(x), slope) for an nn.Conv2d with bias: on CUDA NHWC the convolution runs without its bias and bias +
    activation are one fused pass; otherwise the stock composition"""
    stride = conv.stride if stride is None else stride
    if conv.bias is not None and x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        y0 = F.conv2d(x, conv.weight, None, stride, conv.padding, conv.dilation, conv.groups)
        if bias_leaky_ok(y0, conv.bias):
            return _BiasLeakyReLU.apply(y0, conv.bias.float(), slope)
        return F.leaky_relu(y0 + conv.bias.to(y0.dtype).view(1, -1, 1, 1), slope)
    return F.leaky_relu(F.conv2d(x, conv.weight, conv.bias, stride, conv.padding, conv.dilation, conv.groups), slope)


class _MaxPool3x3s2(Function):
    """nn.MaxPool2d(3, 2, 1) on NHWC activations (csrc/pool.hip): byte-sized argmax map instead of int64 indices, gather
    backward; same values and the same gradient routing (ties included) as ATen"""

    @staticmethod
    def forward(ctx, x):
        L = capi.lib()
        n, c, h, w = x.shape
        y = torch.empty((n, c, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        where = torch.empty(n * (h // 2) * (w // 2) * c, dtype=torch.uint8, device=x.device)
        fn = L.scp_maxpool3x3s2_forward if x.dtype == torch.float32 else L.scp_maxpool3x3s2_forward_bf16
        capi.check(fn(_ptr(x), _ptr(y), _ptr(where), n, h, w, c, capi.current_stream()), "maxpool3x3s2_forward")
        ctx.save_for_backward(where)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = capi.lib()
        (where,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy = _nhwc(dy)
        dx = torch.empty((n, c, h, w), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        fn = L.scp_maxpool3x3s2_backward if dy.dtype == torch.float32 else L.scp_maxpool3x3s2_backward_bf16
        capi.check(fn(_ptr(dy), _ptr(where), _ptr(dx), n, h, w, c, capi.current_stream()), "maxpool3x3s2_backward")
        return dx


def maxpool3x3s2(x, stock):
    """`stock(x)` for stock = nn.MaxPool2d(3, 2, 1): HIP kernels on CUDA NHWC fp32 / bf16 with even H, W; the module otherwise"""
    if (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
            and x.shape[1] % 4 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and stock.kernel_size == 3 and stock.stride == 2 and stock.padding == 1 and stock.dilation == 1 and not stock.ceil_mode):
        return _MaxPool3x3s2.apply(x)
    return stock(x)
