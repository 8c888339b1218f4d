"""tests/posefit_inputs.py -- seeded synthetic inputs for the test-time pose-fitting path (pure torch; shared by
the fixture generator tests/golden/make_golden.py, the tests and bench.py --workload posefit)."""
import torch


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
