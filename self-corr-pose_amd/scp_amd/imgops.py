"""scp_amd/imgops.py -- the three torchvision image ops the training step calls (torchvision is not
on the target image): Normalize, ColorJitter, rotate.

torchvision 0.11 is an un-vendored dependency of the reference (README.md:25-29), so these follow
its published algorithms and are "parity unpinned" against the real package: golden runs replace
the jitter by identity and use angles that are multiples of 90 degrees, where `rotate` is exact
(the same substitution is made on the reference side, tests/golden/ref_harness.py).
Call sites in the reference: model/module/encoder.py:18-19,31 and correspondence.py:87-89.
"""
import ctypes
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from .hostconst import small_tensor


class Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.host_mean, self.host_std = [float(m) for m in mean], [float(v) for v in std]
        self.register_buffer("mean", torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1), persistent=False)

    def forward(self, x):
        return (x - self.mean) / self.std


def _gray(img):
    r, g, b = img.unbind(-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1)


def _rgb_to_hsv(img):
    r, g, b = img.unbind(-3)
    maxc, minc = img.max(-3)[0], img.min(-3)[0]
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), -3)


def _hsv_to_rgb(img):
    h, s, v = img.unbind(-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = (v * (1.0 - s)).clamp(0, 1)
    q = (v * (1.0 - s * f)).clamp(0, 1)
    t = (v * (1.0 - s * (1.0 - f))).clamp(0, 1)
    # sector table (torchvision builds the same table and contracts it with a one-hot einsum, which
    # lands on a degenerate K=6 batched GEMM; a gather is the same selection)
    idx = i.to(torch.int64).unsqueeze(-3)
    r = torch.gather(torch.stack((v, q, p, p, t, v), -3), -3, idx)
    g = torch.gather(torch.stack((t, v, v, q, p, p), -3), -3, idx)
    b = torch.gather(torch.stack((p, p, t, v, v, q), -3), -3, idx)
    return torch.cat((r, g, b), -3)


class ColorJitter(nn.Module):
    """brightness / contrast / saturation / hue jitter, one random order and one factor set per call
    (applied to the whole batch, as torchvision does for a batched tensor)"""

    def __init__(self, brightness=0., contrast=0., saturation=0., hue=0.):
        super().__init__()
        self.brightness, self.contrast, self.saturation, self.hue = brightness, contrast, saturation, hue

    def draw(self):
        """this call's op order and factors: (order, [brightness, contrast, saturation factor], hue shift).
        Host RNG, no device sync."""
        order = torch.randperm(4).tolist()
        rnd = torch.empty(4).uniform_(-1, 1).tolist()
        amount = (self.brightness, self.contrast, self.saturation)
        order = [op for op in order if (amount[op] if op < 3 else self.hue) > 0]
        return order, [1.0 + amount[k] * rnd[k] for k in range(3)], self.hue * rnd[3]

    def forward(self, img):
        return self.apply_ops(img, *self.draw())

    @staticmethod
    def apply_ops(img, order, ratio, hue_shift):
        for op in order:
            if op == 0:
                img = _blend(img, torch.zeros_like(img), ratio[0])
            elif op == 1:
                mean = _gray(img).mean((-3, -2, -1), keepdim=True)
                img = _blend(img, mean, ratio[1])
            elif op == 2:
                img = _blend(img, _gray(img), ratio[2])
            elif op == 3:
                hsv = _rgb_to_hsv(img)
                h = (hsv[..., 0:1, :, :] + hue_shift) % 1.0
                img = _hsv_to_rgb(torch.cat((h, hsv[..., 1:, :, :]), -3))
        return img


def jitter_normalize(img, jitter, normalize, channels_last=True):
    """`normalize(jitter(img))` for the encoder.  CUDA: ONE fused HIP pass that lands the result in the layout its consumer reads
    (channels_last for the NHWC convolutions, NCHW for the own 7x7 stem kernel; csrc/imgops.hip; no fallback -- capi raises without
    the library).  CPU (golden runs, CPU baseline): the torch composition above."""
    if not img.is_cuda or not isinstance(jitter, ColorJitter):
        return normalize(jitter(img))
    from . import capi
    order, ratio, hue_shift = jitter.draw()
    n, c, h, w = img.shape
    if c != 3:
        raise ValueError("jitter_normalize expects RGB images [N,3,H,W]")
    img = img.contiguous().float()
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=img.device,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    L = capi.lib()
    ws = torch.empty(max(1, L.scp_color_jitter_workspace(n) // 4), dtype=torch.float32, device=img.device)
    i4, f3 = ctypes.c_int * 4, ctypes.c_float * 3
    slots = list(order) + [-1] * (4 - len(order))
    capi.check(L.scp_color_jitter_normalize(
        capi.dev_ptr(img, "img"), n, h, w, i4(*slots), f3(*ratio), f3(*[1.0 - r for r in ratio]),
        float(hue_shift), f3(*normalize.host_mean),
        f3(*normalize.host_std), int(bool(channels_last)), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
        ws.numel() * 4, capi.current_stream()), "color_jitter_normalize")
    return out


def rotate(img, angle, interpolation="nearest"):
    """counter-clockwise rotation about the image centre, zero fill, output size = input size.
    Multiples of 90 degrees on square images are exact (rot90); other angles resample through an
    inverse affine grid like torchvision's tensor backend."""
    k = angle / 90.0
    if img.shape[-1] == img.shape[-2] and abs(k - round(k)) < 1e-9:
        return torch.rot90(img, int(round(k)) % 4, dims=(-2, -1))
    squeeze = img.dim() == 3
    if squeeze:
        img = img[None]
    n, _, h, w = img.shape
    a = math.radians(angle)
    # output pixel (x, y), centred, y down -> source location: rotate by +a in image coordinates
    cos, sin = math.cos(a), math.sin(a)
    theta = small_tensor([[cos, -sin * h / w, 0.0], [sin * w / h, cos, 0.0]], img.dtype, img.device)
    grid = F.affine_grid(theta[None].expand(n, -1, -1), (n, 1, h, w), align_corners=False)
    out = F.grid_sample(img, grid, mode="bilinear" if interpolation == "bilinear" else "nearest",
                        padding_mode="zeros", align_corners=False)
    return out[0] if squeeze else out
