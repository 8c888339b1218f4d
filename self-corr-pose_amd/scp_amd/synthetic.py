"""scp_amd/synthetic.py -- seeded synthetic workloads (own code): the prior-mesh stand-ins and the training batch of
SURVEY.md 8(d).  Used by bench.py, __graft_entry__.smoke() and the tests; there is no network for the Wild6D data or the
shipped .obj priors on the GPU box, so throughput runs use these.

  icosphere(3) / bottle_like(3): 642 verts / 1280 faces = the topology of the reference's config/bottle_wild6d/bottle.obj
  (BASELINE's "1280" mesh); subdiv 4: 2562 / 5120 (configs[4]).
  make_batch: img ~ U(0,1); mask = filled ellipse around the image centre; depth = mask * (500 + smooth noise); crop
  intrinsics around f = 755 px (256-px crop), principal point near the centre; returned as the 12-tuple MeshNet.forward
  consumes (after Trainer.batch_reshape's NDC conversion, model/trainer.py:79-103).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


LOOK_AT_Z = 1.0 / math.tan(math.radians(30.0)) + 1.0  # 2.7320508 (SoftRas' default look_at eye)


def octahedron():
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
    f = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4],
                  [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int64)
    return v, f


def icosphere(subdiv=3):
    """Unit icosphere: subdiv 3 -> 642 verts / 1280 faces, subdiv 4 -> 2562 / 5120."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
         [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4],
         [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8],
         [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return np.asarray(v, np.float64), np.asarray(f, np.int64)


def bottle_like(subdiv=3):
    """An icosphere squashed into an elongated, slightly waisted body (642/1280 at subdiv 3):
    same topology as the reference's bottle prior, different geometry."""
    v, f = icosphere(subdiv)
    v = v.copy()
    r = 0.42 + 0.10 * np.cos(2.2 * v[:, 1])
    v[:, 0] *= r
    v[:, 2] *= r
    return v, f


def random_rotations(n, rng):
    out = []
    for _ in range(n):
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        out.append(q)
    return np.stack(out)


def camera_batch(n, rng, depth=5.0, foc=5.9, pp_range=0.2):
    """rotation [n,3,3], translation [n,1,3], foc [n,2], pp [n,2] in NDC units."""
    rot = random_rotations(n, rng)
    trans = np.zeros((n, 1, 3))
    trans[:, 0, 2] = depth + rng.uniform(-0.3, 0.3, n)
    trans[:, 0, :2] = rng.uniform(-0.1, 0.1, (n, 2))
    f = foc * rng.uniform(0.9, 1.1, (n, 2))
    pp = rng.uniform(-pp_range, pp_range, (n, 2))
    return rot, trans, f, pp


def project(verts, rot, trans, foc, pp):
    """verts [n,V,3] (or [V,3]) -> SoftRas-space vertices [n,V,3] float32 (x, y up, z + eye)."""
    if verts.ndim == 2:
        verts = np.broadcast_to(verts, (rot.shape[0],) + verts.shape)
    cam = verts @ rot + trans
    x = pp[:, None, 0] + cam[..., 0] * foc[:, None, 0] / cam[..., 2]
    y = -(pp[:, None, 1] + cam[..., 1] * foc[:, None, 1] / cam[..., 2])
    z = cam[..., 2] + LOOK_AT_Z
    return np.stack([x, y, z], -1).astype(np.float32)


def face_gather(per_vertex, faces):
    """[n,V,C] , [F,3] -> [n,F,3,C]"""
    return np.ascontiguousarray(per_vertex[:, faces])


def raster_inputs(verts, faces, n, seed, tex="rand"):
    """Convenience: (face_vertices [n,F,3,3], face_textures [n,F,3,3]) float32."""
    rng = np.random.default_rng(seed)
    rot, trans, foc, pp = camera_batch(n, rng)
    pv = project(verts, rot, trans, foc, pp)
    fv = face_gather(pv, faces)
    if tex == "rand":
        col = rng.uniform(0, 1, (n, verts.shape[0], 3)).astype(np.float32)
    elif tex == "depth":
        col = pv.copy()
        col[..., 2] -= LOOK_AT_Z
    else:
        col = np.broadcast_to(verts.astype(np.float32), (n,) + verts.shape).copy()
    return fv, face_gather(col, faces)


def make_batch(batch_size, repeat, img_size=256, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    n = batch_size * repeat
    s = img_size
    img = torch.rand(n, 3, s, s, generator=g)
    yy, xx = torch.meshgrid(torch.arange(s, dtype=torch.float32), torch.arange(s, dtype=torch.float32), indexing="ij")
    ax = (85 + 22 * torch.rand(n, generator=g)) * s / 256
    ay = (85 + 22 * torch.rand(n, generator=g)) * s / 256
    c = (s - 1) / 2
    mask = (((xx[None] - c) / ax[:, None, None]) ** 2 + ((yy[None] - c) / ay[:, None, None]) ** 2 <= 1).float()
    noise = F.avg_pool2d(torch.randn(n, 1, s, s, generator=g), 15, 1, 7)[:, 0]
    depth = mask * (500 + 40 * noise * 7)
    foc_px = 755 * s / 256 * (0.9 + 0.2 * torch.rand(n, 2, generator=g))
    pp_px = s / 2 + (50 * torch.rand(n, 2, generator=g) - 25) * s / 256
    pp_crop = pp_px / (s / 2.) - 1.
    foc_crop = foc_px / (s / 2.)
    z2 = torch.zeros(n, 2)
    data = (img, mask, depth, None, z2, z2, foc_px, foc_crop, pp_px, pp_crop, torch.arange(n), None)
    return tuple(t.to(device) if torch.is_tensor(t) else t for t in data)


def posefit_inputs(bsz=3, size=64, n_verts=42, seed=7, outlier_frac=0.15, noise=0.004):
    """Seeded synthetic test-time inputs for Tester.pose_fitting: per pixel a depth map (mm) and a canonical
    coordinate `match` that is the inverse of a known similarity transform of the back-projected point, plus
    noise and a fraction of gross outliers; shared by the generator and by tests (pure torch, no reference)."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    u = (xs.float() + 0.5) / (size / 2) - 1
    v = (ys.float() + 0.5) / (size / 2) - 1
    foc_crop = 5.9 * (0.9 + 0.2 * torch.rand(bsz, 2, generator=g))
    pp_crop = 0.4 * torch.rand(bsz, 2, generator=g) - 0.2
    depth = torch.zeros(bsz, size, size)
    mask = torch.zeros(bsz, size, size)
    conf = torch.zeros(bsz, 1, size, size)
    match = torch.zeros(bsz, 3, size, size)
    truth = []
    for i in range(bsz):
        r = 0.55 + 0.3 * torch.rand(1, generator=g).item()
        inside = ((u / r) ** 2 + (v / (0.8 * r)) ** 2) < 1
        d = 480 + 60 * torch.rand(1, generator=g).item() - 45 * torch.sqrt(torch.clamp(1 - (u / r) ** 2 - (v / (0.8 * r)) ** 2, min=0))
        d = d + 2.0 * torch.randn(size, size, generator=g)
        holes = torch.rand(size, size, generator=g) < 0.05           # missing depth
        depth[i] = torch.where(inside & ~holes, d, torch.zeros_like(d))
        mask[i] = inside.float()
        conf[i, 0] = (torch.rand(size, size, generator=g) > 0.1).float()
        x = (u - pp_crop[i, 0]) / foc_crop[i, 0]
        y = (v - pp_crop[i, 1]) / foc_crop[i, 1]
        pts = torch.stack((x * depth[i], y * depth[i], depth[i]), 0).reshape(3, -1)         # camera points, mm
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        scale = 120 + 60 * torch.rand(1, generator=g).item()
        trans = torch.tensor([8.0, -5.0, 500.0]) + 6 * torch.randn(3, generator=g)
        # target = scale * R^T-ish applied to source (the reference convention is whatever Umeyama recovers);
        # build the source as the exact inverse similarity of the target
        src = (q.T @ (pts - trans[:, None])) / scale
        src = src + noise * torch.randn(src.shape, generator=g)
        out = torch.rand(size * size, generator=g) < outlier_frac
        src[:, out] = 0.6 * torch.randn(3, int(out.sum()), generator=g)
        match[i] = src.reshape(3, size, size)
        truth.append((scale, q, trans))
    pred_v = 0.4 * torch.randn(bsz, n_verts, 3, generator=g)
    return dict(depth=depth, mask=mask, match_conf=conf, match=match, foc_crop=foc_crop, pp_crop=pp_crop, pred_v=pred_v), truth


def umeyama_case(n, outlier_frac, mirror, seed):
    """source ~ unit scale, target = s R source + t (+ small noise), a fraction of targets displaced far
    beyond the pass threshold; `mirror` flips one source axis so the SVD sign correction is exercised"""
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(n, 3, generator=g) * torch.tensor([0.8, 0.5, 0.3])
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    s = 0.8 + 0.8 * torch.rand(1, generator=g).item()
    t = torch.randn(3, generator=g)
    tgt = s * src @ q.T + t + 0.01 * torch.randn(n, 3, generator=g)
    if mirror:
        src = src * torch.tensor([1.0, 1.0, -1.0])
    k = int(outlier_frac * n)
    if k:
        idx = torch.randperm(n, generator=g)[:k]
        tgt[idx] += 8.0 * torch.randn(k, 3, generator=g)
    return src.contiguous(), tgt.contiguous()
