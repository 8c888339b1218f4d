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
