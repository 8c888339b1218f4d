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
