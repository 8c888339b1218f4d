"""oracle/imgops.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of the torchvision 0.11 tensor-backend colour ops the reference's encoder applies
(model/module/encoder.py:18-19,31: `ColorJitter(0.2,0.2,0.2,0.05)` then `Normalize`), following the
published algorithms of torchvision/transforms/functional_tensor.py (adjust_brightness / adjust_contrast /
adjust_saturation / adjust_hue, _blend, _rgb2hsv, _hsv2rgb, rgb_to_grayscale).

PARITY UNPINNED: torchvision is an un-vendored dependency, absent from /root/reference and from this
image, and the reference holds no vectors for these ops -- this file pins the HIP kernel to the published
formulas, not to an executed torchvision."""
import numpy as np

F32 = np.float32


def _gray(img):
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    return (F32(0.2989) * r + F32(0.587) * g + F32(0.114) * b)[:, None]


def _blend(a, b, ratio):
    return np.clip(F32(ratio) * a + F32(1.0 - ratio) * b, 0, 1).astype(F32)


def _rgb2hsv(img):
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    maxc, minc = img.max(1), img.min(1)
    eqc = maxc == minc
    cr = maxc - minc
    ones = np.ones_like(maxc)
    s = cr / np.where(eqc, ones, maxc)
    crd = np.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (F32(2.0) + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (F32(4.0) + gc - rc)
    h = np.fmod((hr + hg + hb) / F32(6.0) + F32(1.0), F32(1.0))
    return np.stack((h, s, maxc), 1).astype(F32)


def _hsv2rgb(img):
    h, s, v = img[:, 0], img[:, 1], img[:, 2]
    i = np.floor(h * F32(6.0))
    f = h * F32(6.0) - i
    i = i.astype(np.int32) % 6
    p = np.clip(v * (F32(1.0) - s), 0, 1)
    q = np.clip(v * (F32(1.0) - s * f), 0, 1)
    t = np.clip(v * (F32(1.0) - s * (F32(1.0) - f)), 0, 1)
    mask = (i[:, None] == np.arange(6).reshape(1, 6, 1, 1)).astype(F32)
    a1 = np.stack((v, q, p, p, t, v), 1)
    a2 = np.stack((t, v, v, q, p, p), 1)
    a3 = np.stack((p, p, t, v, v, q), 1)
    return np.stack(((mask * a1).sum(1), (mask * a2).sum(1), (mask * a3).sum(1)), 1).astype(F32)


def color_jitter_oracle(img, order, ratio, hue_shift):
    """img [N,3,H,W] float32 in [0,1]; order = op ids in application order (0 brightness, 1 contrast,
    2 saturation, 3 hue); ratio[k] = factor of op k"""
    img = np.asarray(img, F32)
    for op in order:
        if op == 0:
            img = _blend(img, np.zeros_like(img), ratio[0])
        elif op == 1:
            mean = _gray(img).mean(axis=(1, 2, 3), keepdims=True, dtype=np.float64).astype(F32)
            img = _blend(img, mean, ratio[1])
        elif op == 2:
            img = _blend(img, _gray(img), ratio[2])
        elif op == 3:
            hsv = _rgb2hsv(img)
            h = np.mod(hsv[:, 0:1] + F32(hue_shift), F32(1.0))
            img = _hsv2rgb(np.concatenate((h, hsv[:, 1:]), 1))
    return img


def normalize_oracle(img, mean, std):
    m = np.asarray(mean, F32).reshape(1, -1, 1, 1)
    s = np.asarray(std, F32).reshape(1, -1, 1, 1)
    return ((img - m) / s).astype(F32)


# ---- torchvision 0.11 `rotate` on tensors (the reference's rotation-cycle augmentation: model/module/correspondence.py:82-89 calls
# torchvision.transforms.functional.rotate(img, angle) with the default NEAREST for masks / grids and BILINEAR for images) -------------
# Published algorithm restated (torchvision/transforms/functional.py `rotate` -> `_get_inverse_affine_matrix(center=[0,0], -angle, ...)`,
# functional_tensor.py `_gen_affine_grid`, `_apply_grid_transform` -> torch.nn.functional.grid_sample(..., padding_mode="zeros",
# align_corners=False)).  PARITY UNPINNED like the colour ops above: no executed torchvision, no recorded vector.
def rotate_oracle(img, angle, interpolation="nearest"):
    """img [N,C,H,W] float32 -> rotated by `angle` degrees counter-clockwise about the image centre, same size, zero fill"""
    import math
    img = np.asarray(img, F32)
    n, c, h, w = img.shape
    rot = math.radians(-angle)
    # _get_inverse_affine_matrix with centre 0, no translation / scale / shear: [d, -b, 0, -c, a, 0] of the forward matrix
    a_, b_, c_, d_ = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    theta = np.array([[d_, -b_, 0.0], [-c_, a_, 0.0]], F32)
    # _gen_affine_grid: pixel-centre coordinates, theta^T rescaled by the half sizes, one bmm
    xs = np.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, w, dtype=F32)
    ys = np.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, h, dtype=F32)
    base = np.stack((np.broadcast_to(xs[None, :], (h, w)), np.broadcast_to(ys[:, None], (h, w)), np.ones((h, w), F32)), -1).reshape(-1, 3)
    rescaled = (theta.T / np.array([0.5 * w, 0.5 * h], F32)).astype(F32)
    grid = (base @ rescaled).astype(F32).reshape(h, w, 2)
    # grid_sample, align_corners=False: normalised -> pixel index space
    ix = ((grid[..., 0] + 1) * w - 1) / 2
    iy = ((grid[..., 1] + 1) * h - 1) / 2
    out = np.zeros_like(img)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = img[:, :, np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return np.where(ok[None, None], v, F32(0))
    if interpolation == "nearest":
        out = tap(np.rint(iy).astype(np.int64), np.rint(ix).astype(np.int64))       # nearbyint: ties to even, like ATen's kernel
    else:
        x0, y0 = np.floor(ix), np.floor(iy)
        fx, fy = (ix - x0).astype(F32), (iy - y0).astype(F32)
        x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
        out = (tap(y0, x0) * ((1 - fx) * (1 - fy)) + tap(y0, x0 + 1) * (fx * (1 - fy)) +
               tap(y0 + 1, x0) * ((1 - fx) * fy) + tap(y0 + 1, x0 + 1) * (fx * fy))
    return out.astype(F32), (ix, iy)
