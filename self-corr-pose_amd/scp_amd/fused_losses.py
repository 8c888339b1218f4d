"""scp_amd/fused_losses.py -- the step's per-pixel image losses as single autograd ops over csrc/losses.hip.

`depth_group_losses(depth_out, match_out, match, depth, mask)` == (compute_mask_loss, compute_depth_loss, compute_match_loss) of
scp_amd/losses.py (model/util/loss_utils.py:236-244, :273-284, :317-320) on the renders as Renderer.render_depth_group_raw holds them;
`texture_loss(tex_out, img, mask)` == compute_texture_loss (:246-252).  Two forward and three backward launches (one + one for the
texture term) instead of ~150 ATen kernels on the step's serial stretch.  The torch compositions in losses.py stay the definition:
tests/test_fused_losses.py compares values and gradients with them in float64; MeshNet uses these ops when `covers()` holds
(CUDA fp32, no occlusion weight, depth loss on, mask shared with the depth pass) and the compositions otherwise.
"""
import ctypes

import torch
from torch.autograd import Function

from . import capi


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def covers(*tensors):
    w = tensors[0].shape[-1]
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors) and 32 <= w <= 1024 and (w & (w - 1)) == 0


def _dense(t, name):
    if not t.is_contiguous():
        t = t.contiguous()
    capi.dev_ptr(t, name)
    return t


class _DepthGroupLosses(Function):
    @staticmethod
    def forward(ctx, depth_out, match, match_out, depth, mask):
        L = capi.lib()
        depth_out, match, match_out = _dense(depth_out, "depth_out"), _dense(match, "match"), _dense(match_out, "match_out")
        depth, mask = _dense(depth, "depth"), _dense(mask, "mask")
        b, _, h, w = depth_out.shape
        if match.shape != (b, 3, h, w) or match_out.shape != (b, 4, h, w) or depth.shape != (b, h, w) or mask.shape != (b, h, w):
            raise RuntimeError("depth_group_losses: shapes %s %s %s %s %s" % tuple(tuple(t.shape) for t in
                                                                                   (depth_out, match, match_out, depth, mask)))
        parts = torch.empty(L.scp_image_losses_parts() * 4, dtype=torch.float32, device=depth_out.device)
        rowsum = torch.empty(b, h, 3, dtype=torch.float32, device=depth_out.device)
        capi.check(L.scp_image_losses_forward(_ptr(depth_out), _ptr(depth), _ptr(mask), _ptr(match), _ptr(match_out), b, h, w,
                                              _ptr(parts), _ptr(rowsum), capi.current_stream()), "image_losses_forward")
        ctx.save_for_backward(depth_out, match, match_out, depth, mask, parts)
        losses = rowsum.mean(1)                                                     # [B,3]
        return 0.2 * losses[:, 0], losses[:, 1], losses[:, 2]

    @staticmethod
    def backward(ctx, g_mask, g_depth, g_match):
        L = capi.lib()
        depth_out, match, match_out, depth, mask, parts = ctx.saved_tensors
        b, _, h, w = depth_out.shape
        zeros = None

        def grad_vec(g):
            nonlocal zeros
            if g is None:
                if zeros is None:
                    zeros = torch.zeros(b, dtype=torch.float32, device=depth_out.device)
                return zeros
            return g.contiguous()
        g_mask, g_depth, g_match = grad_vec(g_mask), grad_vec(g_depth), grad_vec(g_match)
        grad_depth_out = torch.empty_like(depth_out)
        grad_match = torch.empty_like(match)
        gsum = torch.empty(b * h, dtype=torch.float32, device=depth_out.device)
        st = capi.current_stream()
        capi.check(L.scp_image_losses_backward(_ptr(depth_out), _ptr(depth), _ptr(mask), _ptr(match), _ptr(match_out), _ptr(parts),
                                               _ptr(g_mask), _ptr(g_depth), _ptr(g_match), b, h, w, _ptr(grad_depth_out),
                                               _ptr(grad_match), _ptr(gsum), st), "image_losses_backward")
        total = gsum.sum().reshape(1)
        capi.check(L.scp_image_losses_backward_scale(_ptr(depth_out), _ptr(parts), _ptr(total), b, h, w, _ptr(grad_depth_out), st),
                   "image_losses_backward_scale")
        return (grad_depth_out if ctx.needs_input_grad[0] else None, grad_match if ctx.needs_input_grad[1] else None,
                None, None, None)


class _TextureLoss(Function):
    @staticmethod
    def forward(ctx, tex_out, img, mask):
        L = capi.lib()
        tex_out, img, mask = _dense(tex_out, "tex_out"), _dense(img, "img"), _dense(mask, "mask")
        b, _, h, w = tex_out.shape
        if tex_out.shape[1] != 4 or img.shape != (b, 3, h, w) or mask.shape != (b, h, w):
            raise RuntimeError("texture_loss: shapes %s %s %s" % (tuple(tex_out.shape), tuple(img.shape), tuple(mask.shape)))
        rowsum = torch.empty(b, h, dtype=torch.float32, device=tex_out.device)
        capi.check(L.scp_texture_loss_forward(_ptr(tex_out), _ptr(img), _ptr(mask), b, h, w, _ptr(rowsum), capi.current_stream()),
                   "texture_loss_forward")
        ctx.save_for_backward(tex_out, img, mask)
        return rowsum.mean(1)

    @staticmethod
    def backward(ctx, g):
        L = capi.lib()
        tex_out, img, mask = ctx.saved_tensors
        b, _, h, w = tex_out.shape
        grad = torch.empty_like(tex_out)
        capi.check(L.scp_texture_loss_backward(_ptr(tex_out), _ptr(img), _ptr(mask), _ptr(g.contiguous()), b, h, w, _ptr(grad),
                                               capi.current_stream()), "texture_loss_backward")
        return grad, None, None


def depth_group_losses(depth_out, match_out, match, depth, mask):
    """(mask_loss [B], depth_loss [B], match_loss [B]); depth_out [B,4,H,W] (depth plane 2, alpha plane 3), match_out [B,4,H,W]
    (canonical xyz + alpha, no gradient), match [B,3,H,W], depth / mask [B,H,W]"""
    return _DepthGroupLosses.apply(depth_out, match, match_out.detach(), depth, mask)


def texture_loss(tex_out, img, mask):
    """texture loss [B]; tex_out [B,4,H,W] = rgb + alpha of the soft-texture pass"""
    return _TextureLoss.apply(tex_out, img, mask)
