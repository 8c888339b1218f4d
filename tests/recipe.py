"""tests/recipe.py -- seeded synthetic weights, keyed by parameter NAME.

There is no network on the build/GPU machines, so neither the ImageNet ResNet18 nor the DINO
checkpoint exist; parity runs load the same recipe into the reference model (golden generator) and
into the build's model (tests).  Every tensor is drawn from its own generator seeded with
crc32(name), so the two state dicts agree wherever the names agree, independent of key order."""
import math
import zlib

import torch

KEEP = ("mesh.mean_v", "mesh.faces", "mesh.symm_rots", "laplacian", "x_offset", "y_offset", "v0s", "v1s", "v2s", "v3s")


def tensor_for(name, like, seed=1234):
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
    shape = tuple(like.shape)
    if name.endswith("num_batches_tracked") or not like.dtype.is_floating_point:
        return like.clone()
    leaf = name.rsplit(".", 1)[-1]
    is_norm = any(t in name for t in (".bn", "norm", "downsample.1"))
    if leaf == "running_var":
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if is_norm and leaf == "weight":
        return 1.0 + 0.05 * torch.randn(shape, generator=g)
    if is_norm and leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf in ("pos_embed", "cls_token"):
        return 0.02 * torch.randn(shape, generator=g)
    if like.dim() >= 2:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return torch.randn(shape, generator=g) * math.sqrt(1.0 / fan_in)
    return 0.02 * torch.randn(shape, generator=g)


def recipe_state_dict(model, seed=1234, prefix=""):
    out = {}
    for name, t in model.state_dict().items():
        full = prefix + name
        if any(k in full for k in KEEP):
            continue
        out[name] = tensor_for(full, t, seed)
    return out


def load_recipe(model, seed=1234):
    missing, unexpected = model.load_state_dict(recipe_state_dict(model, seed), strict=False)
    assert not unexpected
    return model


def symmetry_sample(n, num_points, num_faces, seed=99):
    """(face_idx [n,P] int64, bary [n,P,3] fp32): a fixed surface sample injected on both sides of a
    parity run in place of pytorch3d's RNG-driven sample_points_from_meshes"""
    g = torch.Generator().manual_seed(seed)
    face_idx = torch.randint(0, num_faces, (n, num_points), generator=g)
    su = torch.rand(n, num_points, generator=g).sqrt()
    r2 = torch.rand(n, num_points, generator=g)
    return face_idx, torch.stack((1.0 - su, su * (1.0 - r2), su * r2), -1)
