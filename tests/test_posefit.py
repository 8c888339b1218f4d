"""test-time pose fitting: oracle vs vectors recorded from the reference's umeyama.py /
Tester.pose_fitting; HIP batched RANSAC-Umeyama vs those vectors and vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import posefit as oracle_posefit

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Stream:
    """replays a recorded RandIdx stream"""

    def __init__(self, rows):
        self.rows, self.pos = np.asarray(rows).reshape(-1, 5), 0

    def __call__(self, n):
        r = self.rows[self.pos]
        self.pos += 1
        assert r.max() < n
        return r


def _close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= tol * (1 + np.abs(b).max()), "%s: max err %.3g" % (what, err)


def test_oracle_umeyama_cases_match_reference():
    d = np.load(os.path.join(GOLD, "umeyama_cases.npz"))
    for c in range(int(d["n_cases"])):
        st = Stream(d["c%d_rand_idx" % c])
        sc, rot, tr, T = oracle_posefit.estimate_similarity_oracle(d["c%d_source" % c], d["c%d_target" % c], st)
        assert st.pos == d["c%d_rand_idx" % c].shape[0]
        _close(sc, d["c%d_scale" % c], 1e-4, "scale")
        _close(rot, d["c%d_rotation" % c], 1e-4, "rotation")
        _close(tr, d["c%d_translation" % c], 1e-4, "translation")
        _close(T, d["c%d_transform" % c], 1e-4, "transform")


def test_oracle_pose_fitting_matches_reference():
    d = np.load(os.path.join(GOLD, "posefit_b3_64.npz"))
    st = Stream(d["rand_idx"])
    bbox, verts, rot, tr = oracle_posefit.pose_fitting_oracle(d["depth"], d["mask"], d["match"], d["match_conf"], d["foc_crop"],
                                                              d["pp_crop"], d["pred_v"], d["base_rot"], st)
    _close(rot, d["rotation"], 1e-4, "rotation")
    _close(tr, d["translation"], 1e-4, "translation")
    _close(bbox, d["bbox"], 1e-4, "bbox")
    _close(verts, d["verts"], 1e-4, "verts")


def test_oracle_rng_stream_is_torch_randint():
    """the recorded stream is what torch.randint(0, n, (5,)) yields after manual_seed(seed): the product draws
    (K,5) blocks from the same generator"""
    d = np.load(os.path.join(GOLD, "umeyama_cases.npz"))
    torch.manual_seed(int(d["c0_seed"]))
    n = d["c0_source"].shape[0]
    block = torch.randint(0, n, (d["c0_rand_idx"].shape[0], 5)).numpy()
    assert np.array_equal(block, d["c0_rand_idx"])


@pytest.mark.gpu
def test_hip_umeyama_cases_match_reference():
    from scp_amd import pose_fit
    d = np.load(os.path.join(GOLD, "umeyama_cases.npz"))
    for c in range(int(d["n_cases"])):
        torch.manual_seed(int(d["c%d_seed" % c]))
        src = torch.from_numpy(d["c%d_source" % c]).cuda()
        tgt = torch.from_numpy(d["c%d_target" % c]).cuda()
        sc, rot, tr, T = pose_fit.estimateSimilarityTransform(src, tgt)
        assert pose_fit.last_report["rounds"] == [d["c%d_rand_idx" % c].shape[0]]
        assert pose_fit.last_report["n_inliers"] == [int(d["c%d_n_inliers" % c])]
        _close(sc.cpu(), d["c%d_scale" % c], 1e-4, "scale")
        _close(rot.cpu(), d["c%d_rotation" % c], 1e-4, "rotation")
        _close(tr.cpu().reshape(-1), d["c%d_translation" % c], 1e-4, "translation")
        _close(T.cpu(), d["c%d_transform" % c], 1e-4, "transform")


@pytest.mark.gpu
def test_hip_pose_fitting_matches_reference():
    from scp_amd import pose_fit
    d = np.load(os.path.join(GOLD, "posefit_b3_64.npz"))
    t = {k: torch.from_numpy(d[k]).cuda() for k in ("depth", "mask", "match", "match_conf", "foc_crop", "pp_crop", "pred_v", "base_rot")}
    torch.manual_seed(int(d["seed"]))
    fit = pose_fit.PoseFitter(img_size=d["depth"].shape[-1], base_rot=t["base_rot"])
    bbox, verts, rot, tr = fit.pose_fitting(t["depth"], t["mask"], t["match"], t["match_conf"], t["foc_crop"], t["pp_crop"], t["pred_v"])
    assert pose_fit.last_report["n_points"] == d["n_points"].tolist()
    _close(rot.cpu(), d["rotation"], 1e-4, "rotation")
    _close(tr.cpu(), d["translation"], 1e-4, "translation")
    _close(bbox.cpu(), d["bbox"], 1e-4, "bbox")
    _close(verts.cpu(), d["verts"], 1e-4, "verts")
    # the generator is left where the reference leaves it
    torch.manual_seed(int(d["seed"]))
    for n in d["n_points"]:
        torch.randint(0, int(n), (d["rand_idx"].shape[1], 5))
    expect = torch.get_rng_state()
    torch.manual_seed(int(d["seed"]))
    fit.pose_fitting(t["depth"], t["mask"], t["match"], t["match_conf"], t["foc_crop"], t["pp_crop"], t["pred_v"])
    assert torch.equal(torch.get_rng_state(), expect)


@pytest.mark.gpu
def test_failed_frame_gets_default_pose_and_the_rest_is_unchanged():
    """tester.py:369-379: a frame whose fit fails (no valid correspondence: torch.randint(0, 0) raises in the reference)
    gets scale 100 / R = I / t = [0,0,500] and evaluation continues; the other frames and the generator state are exactly
    what they are without the empty frame in between (it consumes no random numbers)"""
    from scp_amd import pose_fit
    d = np.load(os.path.join(GOLD, "posefit_b3_64.npz"))
    keys = ("depth", "mask", "match", "match_conf", "foc_crop", "pp_crop", "pred_v")
    t = {k: torch.from_numpy(d[k]).cuda() for k in keys + ("base_rot",)}
    fit = pose_fit.PoseFitter(img_size=d["depth"].shape[-1], base_rot=t["base_rot"])
    torch.manual_seed(int(d["seed"]))
    clean = fit.pose_fitting(*(t[k] for k in keys))
    state_clean = torch.get_rng_state()
    # insert an image with an empty mask (and one with 3 valid pixels) between images 0 and 1
    def widen(x, fill):
        return torch.cat((x[:1], fill(x[:1]), fill(x[:1]), x[1:]), 0)
    w = {k: widen(t[k], lambda a: a.clone()) for k in keys}
    w["mask"][1] = 0
    w["mask"][2] = 0
    valid = ((w["depth"][2] > 0) & (w["match_conf"][2, 0] > 0)).nonzero()[:3]
    w["mask"][2][valid[:, 0], valid[:, 1]] = 1
    torch.manual_seed(int(d["seed"]))
    bbox, verts, rot, tr = fit.pose_fitting(*(w[k] for k in keys))
    assert pose_fit.last_report["failed"] == [1, 2]
    assert torch.equal(torch.get_rng_state(), state_clean)
    keep = [0, 3, 4]
    for got, ref in zip((bbox, verts, rot, tr), clean):
        assert torch.equal(got[keep], ref)
    base = t["base_rot"].reshape(3, 3)
    for b in (1, 2):
        assert torch.allclose(rot[b], base)                               # base_rot @ I
        assert torch.allclose(tr[b].reshape(-1), torch.tensor([0., 0., 0.5], device="cuda"))   # 500 mm -> m
    with pytest.raises(RuntimeError):
        src, tgt, counts = fit.correspondences(*(w[k].float() for k in keys[:-1]))
        pose_fit.fit_padded(src, tgt, counts, strict=True)


@pytest.mark.gpu
def test_hip_pose_fitting_full_size_against_oracle():
    """bench geometry (B=8 here, 256x256): HIP vs the numpy oracle fed the same index stream"""
    from posefit_inputs import posefit_inputs
    from scp_amd import pose_fit
    data, _ = posefit_inputs(bsz=8, size=256, n_verts=642, seed=3)
    base = torch.eye(3)[None]
    torch.manual_seed(5)
    fit = pose_fit.PoseFitter(img_size=256, base_rot=base.cuda())
    out = fit.pose_fitting(*(data[k].cuda() for k in ("depth", "mask", "match", "match_conf", "foc_crop", "pp_crop", "pred_v")))
    torch.manual_seed(5)
    ref = oracle_posefit.pose_fitting_oracle(*(data[k].numpy() for k in ("depth", "mask", "match", "match_conf", "foc_crop", "pp_crop", "pred_v")),
                                             base.numpy(), lambda n: torch.randint(0, n, (5,)).numpy())
    for a, b, what in zip(out, ref, ("bbox", "verts", "rotation", "translation")):
        _close(a.cpu(), b, 2e-4, what)


def test_sequential_selection_rule():
    from scp_amd.pose_fit import _select
    res = np.array([5, 7, 3, 3, 2, 9], np.float32)
    assert _select(res, np.float32(0.5)) == (4, 6)          # no stop: first strictly-smaller wins, all rounds run
    assert _select(res, np.float32(3.5)) == (2, 3)          # best drops under the stop threshold at round 3
    assert _select(res, np.float32(100)) == (0, 1)


@pytest.mark.gpu
def test_early_stop_rewinds_generator(monkeypatch):
    """The reference's 5-point fits are biased (unbiased variance over 5 points, umeyama.py:189-190), so its stop
    criterion practically never fires on real data; force it with a large StopThreshold and check that problems
    then consume the generator round by round exactly like the sequential loop: problem b+1's indices are drawn
    after problem b's LAST EXECUTED round."""
    from scp_amd import pose_fit
    g = torch.Generator().manual_seed(0)
    counts = [150, 200, 120]
    src = torch.randn(3, 200, 3, generator=g)
    tgt = 1.2 * src + 0.05 * torch.randn(3, 200, 3, generator=g)
    real = pose_fit._thresholds

    def patched(source, target, counts_dev):
        p, _ = real(source, target, counts_dev)
        return p, torch.full_like(p, 1e9)                       # every first round "converges"

    monkeypatch.setattr(pose_fit, "_thresholds", patched)
    torch.manual_seed(21)
    scale, rot, tr, T, _ = pose_fit.fit_padded(src.cuda(), tgt.cuda(), counts)
    after = torch.get_rng_state()
    assert pose_fit.last_report["rounds"] == [1, 1, 1] and pose_fit.last_report["chosen"] == [0, 0, 0]
    torch.manual_seed(21)
    for b, n in enumerate(counts):
        S = np.concatenate((src[b, :n].numpy(), np.ones((n, 1), np.float32)), 1).T
        Tg = np.concatenate((tgt[b, :n].numpy(), np.ones((n, 1), np.float32)), 1).T
        tn = np.linalg.norm(tgt[b, :n].numpy(), axis=1).mean()
        sn = np.linalg.norm(src[b, :n].numpy(), axis=1).mean()
        s_in, t_in, _, rounds = oracle_posefit.ransac_oracle(S, Tg, lambda m: torch.randint(0, m, (5,)).numpy(), 100,
                                                             np.float32(max(tn / sn, sn / tn)), np.float32(1e9))
        assert rounds == 1
        _close(T[b].cpu(), oracle_posefit.umeyama_oracle(s_in, t_in)[3], 1e-4, "transform %d" % b)
    assert torch.equal(after, torch.get_rng_state())


@pytest.mark.gpu
@pytest.mark.child_process
def test_tester_predict_end_to_end():
    """eval-mode MeshNet forward -> batched pose fitting on the GPU at 256x256 (recipe weights): shapes, finiteness,
    proper rotations, and the fitted similarity reproduces the correspondences it was fitted to"""
    import recipe
    import scenes
    import synth
    import scp_amd.dino as dino
    from scp_amd import tester
    from scp_amd.flags import Options
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=2, repeat=2, train=False)
    t = tester.Tester(opts, prior=scenes.bottle_like(3))
    model = t.define_model()
    recipe.load_recipe(model)
    data = synth.make_batch(2, 2, opts.img_size, seed=5, device="cuda")
    torch.manual_seed(0)
    pred, (bbox, verts, rot, tr) = t.predict(data)
    bsz = data[0].shape[0]
    assert bbox.shape == (bsz, 9, 3) and rot.shape == (bsz, 3, 3) and tr.shape == (bsz, 1, 3) and verts.shape == pred[0].shape
    for x in (bbox, verts, rot, tr):
        assert torch.isfinite(x).all()
    base = tester.get_base_rot(opts, "cuda")
    r = base.transpose(1, 2) @ rot                       # undo base_rot: the fitted Rotation itself
    eye = torch.eye(3, device="cuda")[None]
    assert (r @ r.transpose(1, 2) - eye).abs().max() < 1e-4 and (torch.linalg.det(r) - 1).abs().max() < 1e-4
    out = t.eval_deg_cm((bbox, verts, rot, tr), (rot.transpose(1, 2).cpu(), bbox[:, 0].cpu(), torch.ones(bsz, 3)))
    # identical poses: the reference's unclipped arccos((tr - 1) / 2) sits at the edge of its domain, NaN allowed
    assert len(out) == bsz and all((np.isnan(a) or a < 0.1) and c < 1e-3 for a, c in out)


def test_deg_cm_metric_matches_reference():
    """scp_amd.tester.get_best_deg_cm vs eval_utils.get_best_deg_cm + the vendored objectron Box, recorded on the fitted
    boxes of the pose-fitting fixture (tests/golden/posefit_metric.npz)"""
    from scp_amd.tester import get_best_deg_cm
    d = np.load(os.path.join(GOLD, "posefit_metric.npz"))
    for i in range(int(d["n"])):
        for sym in (0, 1):
            ang, cm = get_best_deg_cm(sym, d["bbox"][i].astype(np.float64), d["rotation"][i].T.astype(np.float64),
                                      d["gt%d_rot" % i].astype(np.float64), d["gt%d_trans" % i].astype(np.float64),
                                      d["gt%d_scale" % i].astype(np.float64))
            ref = d["gt%d_sym%d" % (i, sym)]
            assert abs(cm - ref[1]) < 1e-6 * (1 + ref[1])
            # near 0 the unclipped arccos turns 1e-7 of rounding in the matrix into ~0.01 degrees (thresholds are 5 / 10)
            assert abs(ang - ref[0]) < 2e-2 + 1e-4 * ref[0], (i, sym, ang, ref[0])


def test_box_iou_matches_reference():
    """scp_amd.eval_nocs.get_best_iou vs eval_utils.get_best_iou over the vendored objectron exact IoU (recorded): partial
    overlaps, identical boxes, disjoint boxes, with and without the 18-fold y-symmetry search"""
    from scp_amd import eval_nocs
    d = np.load(os.path.join(GOLD, "posefit_metric.npz"))
    for i in range(int(d["n"])):
        for j in range(4):
            for k, sym in enumerate((0, 1)):
                got = eval_nocs.get_best_iou(sym, d["bbox"][i].astype(np.float64), d["iou%d_%d_rot" % (i, j)],
                                             d["iou%d_%d_trans" % (i, j)], d["iou%d_%d_scale" % (i, j)])
                ref = float(d["iou%d_%d" % (i, j)][k])
                assert abs(got - ref) < 2e-6 + 1e-5 * ref, (i, j, sym, got, ref)


def test_box_iou_closed_forms():
    from scp_amd import eval_nocs
    a = eval_nocs.box_from_transformation(np.eye(3), np.zeros(3), np.array([2.0, 2.0, 2.0]))
    b = eval_nocs.box_from_transformation(np.eye(3), np.array([1.0, 0.0, 0.0]), np.array([2.0, 2.0, 2.0]))
    assert abs(eval_nocs.box_iou(a, b) - (4.0 / 12.0)) < 1e-12            # overlap 1x2x2 of two 8-volume cubes
    c = eval_nocs.box_from_transformation(np.eye(3), np.array([3.0, 0.0, 0.0]), np.array([2.0, 2.0, 2.0]))
    assert eval_nocs.box_iou(a, c) == 0.0
    th = np.pi / 4
    rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    r = eval_nocs.box_from_transformation(rz, np.zeros(3), np.array([2.0, 2.0, 2.0]))
    inter = 2.0 * (8.0 * (np.sqrt(2.0) - 1.0))                            # regular octagon of the two squares x height 2
    assert abs(eval_nocs.box_iou(a, r) - inter / (16.0 - inter)) < 1e-9
    rot, trans, scale = eval_nocs.box_fit(r)
    assert np.allclose(rot, rz) and np.allclose(scale, 2.0) and np.allclose(trans, 0.0)
