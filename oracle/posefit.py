"""oracle/posefit.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy (float32) restatement of the reference's test-time pose fitting:
  model/util/umeyama.py:9-38    estimateSimilarityTransform (threshold heuristics, failure on < 10 % inliers)
  model/util/umeyama.py:97-121  getRANSACInliers (100 rounds, 5 random pairs, first-best-wins, early stop)
  model/util/umeyama.py:123-131 evaluateModel
  model/util/umeyama.py:161-201 estimateSimilarityUmeyama
  model/tester.py:324-427       Tester.pose_fitting (mask, back-projection, per-image fit, box)
Pinned against tests/golden/posefit_b3_64.npz and umeyama_cases.npz, which were recorded by running the
reference functions themselves (tests/golden/make_golden.py gen_posefit), including the RandIdx stream.
`draw(n)` supplies the 5 indices of a round (recorded stream, or torch.randint(0, n, (5,)))."""
import numpy as np

F32 = np.float32


def umeyama_oracle(SourceHom, TargetHom):
    src, tgt = SourceHom[:3].astype(F32), TargetHom[:3].astype(F32)
    n = src.shape[1]
    sc, tc = src.mean(1, dtype=F32), tgt.mean(1, dtype=F32)
    cs, ct = src - sc[:, None], tgt - tc[:, None]
    cov = (ct @ cs.T) / F32(n)
    if np.isnan(cov).any():
        raise RuntimeError("There are NANs in the input.")
    U, D, Vh = np.linalg.svd(cov, full_matrices=True)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0.0:
        D = D.copy(); U = U.copy()
        D[-1] = -D[-1]
        U[:, -1] = -U[:, -1]
    rotation = (U @ Vh).T
    varp = src.var(axis=1, ddof=1, dtype=F32).sum(dtype=F32)
    scale_fact = F32(1) / varp * D.sum(dtype=F32)
    scales = np.repeat(scale_fact, 3).astype(F32)
    translation = tc - (sc[None] @ (scale_fact * rotation))[0]
    out = np.eye(4, dtype=F32)
    out[:3, :3] = np.diag(scales) @ rotation
    out[:3, 3] = translation
    return scales, rotation.astype(F32), translation.astype(F32), out


def evaluate_model_oracle(OutTransform, SourceHom, TargetHom, PassThreshold):
    diff = TargetHom - OutTransform @ SourceHom
    rvec = np.linalg.norm(diff[:3], axis=0).astype(F32)
    residual = F32(np.linalg.norm(rvec))
    idx = np.nonzero(rvec < PassThreshold)[0]
    return residual, idx.shape[0] / SourceHom.shape[1], idx


def ransac_oracle(SourceHom, TargetHom, draw, MaxIterations=100, PassThreshold=200, StopThreshold=1):
    best_res, best_ratio = 1e10, 0
    best_idx = np.arange(SourceHom.shape[1])
    rounds = 0
    for _ in range(MaxIterations):
        rounds += 1
        r = np.asarray(draw(SourceHom.shape[1]))
        _, _, _, T = umeyama_oracle(SourceHom[:, r], TargetHom[:, r])
        res, ratio, idx = evaluate_model_oracle(T, SourceHom, TargetHom, PassThreshold)
        if res < best_res:
            best_res, best_ratio, best_idx = res, ratio, idx
        if best_res < StopThreshold:
            break
    return SourceHom[:, best_idx], TargetHom[:, best_idx], best_ratio, rounds


def estimate_similarity_oracle(source, target, draw):
    source, target = np.asarray(source, F32), np.asarray(target, F32)
    ones = np.ones((source.shape[0], 1), F32)
    SourceHom = np.concatenate((source, ones), 1).T
    TargetHom = np.concatenate((target, ones), 1).T
    tn = np.linalg.norm(target, axis=1).astype(F32).mean(dtype=F32)
    sn = np.linalg.norm(source, axis=1).astype(F32).mean(dtype=F32)
    ts, st = tn / sn, sn / tn
    pass_t = st if st > ts else ts
    stop_t = pass_t / F32(100)
    s_in, t_in, ratio, rounds = ransac_oracle(SourceHom, TargetHom, draw, 100, pass_t, stop_t)
    if ratio < 0.1:
        return None, None, None, None
    return umeyama_oracle(s_in, t_in)


def pose_fitting_oracle(depth, mask, match, match_conf, foc_crop, pp_crop, pred_v, base_rot, draw):
    """tester.py:324-427 -> bbox [B,9,3], verts [B,V,3], rotation [B,3,3], translation [B,1,3]"""
    depth, mask, match, match_conf = (np.asarray(a, F32) for a in (depth, mask, match, match_conf))
    foc_crop, pp_crop, pred_v, base_rot = (np.asarray(a, F32) for a in (foc_crop, pp_crop, pred_v, base_rot))
    bsz, h, w = depth.shape
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    grid = (np.stack((xs, ys)).reshape(2, -1).astype(F32) + F32(0.5)) / F32(w / 2) - F32(1)   # tester.py:134-135
    valid = ((depth > 0)[:, None] * mask[:, None] * match_conf).reshape(bsz, -1) > 0
    rots, transs, scales = [], [], []
    for i in range(bsz):
        intr = np.eye(3, dtype=F32)
        intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2] = foc_crop[i, 0], foc_crop[i, 1], pp_crop[i, 0], pp_crop[i, 1]
        intr_inv = np.linalg.inv(intr).astype(F32)
        v = valid[i]
        uv = grid[:, v].T
        m = match[i].reshape(3, -1)[:, v].T
        d = depth[i].reshape(-1)[v][:, None]
        xyz = np.concatenate((uv, np.ones((uv.shape[0], 1), F32)), 1) @ intr_inv.T
        pts = xyz * d / xyz[:, 2:]
        sc, rot, tr, _ = estimate_similarity_oracle(m, pts, draw)
        rots.append(rot.reshape(3, 3)); transs.append(tr.reshape(-1)); scales.append(sc.reshape(-1))
    rotation = np.stack(rots)
    translation = (np.stack(transs) * F32(0.001)).reshape(-1, 1, 3)
    scale_fit = (np.stack(scales) * F32(0.001)).reshape(-1, 1, 3)
    br = np.repeat(base_rot.reshape(1, 3, 3), bsz, 0)
    pv = pred_v @ br.transpose(0, 2, 1)
    rotation = br @ rotation
    lo, hi = pv.min(1), pv.max(1)
    corners = [(lo + hi) / 2]
    for sx in (0, 1):
        for sy in (0, 1):
            for sz in (0, 1):
                corners.append(np.stack(((hi if sx else lo)[:, 0], (hi if sy else lo)[:, 1], (hi if sz else lo)[:, 2]), -1))
    bbox = np.stack(corners, 1)
    bbox = (bbox * scale_fit) @ rotation + translation
    verts = (pv * scale_fit) @ rotation + translation
    return bbox.astype(F32), verts.astype(F32), rotation.astype(F32), translation.astype(F32)
