"""G6: scp_amd.losses / Correspondence eval branch vs the golden vectors recorded from the reference's
model/util/loss_utils.py and correspondence.py:58-69 (values and gradients, fp32 round-off)."""
import numpy as np
import torch

import golden_io


def _t(a, grad=False):
    return torch.tensor(a, requires_grad=grad)


def test_image_space_losses_match_reference():
    from scp_amd import losses as L
    d = golden_io.load("losses_small")
    img, mask = _t(d["img"]), _t(d["mask"])
    mask_pred, tex_pred, tex_mask = _t(d["mask_pred"], True), _t(d["tex_pred"], True), _t(d["tex_mask"], True)
    depth_pred, match, imatch = _t(d["depth_pred"], True), _t(d["match"], True), _t(d["imatch"], True)
    m = L.compute_mask_loss(img, mask, mask_pred)
    t = L.compute_texture_loss(img, mask, tex_pred, tex_mask)
    dl, ddiff = L.compute_depth_loss(_t(d["depth"]), depth_pred, _t(d["depth_mask"]), mask)
    ml = L.compute_match_loss(match, _t(d["match_gt"]), _t(d["match_mask"]), mask)
    il = L.compute_imatch_loss(imatch, _t(d["imatch_gt"]), _t(d["depth_weight"]))
    for got, key in ((m, "mask_loss"), (t, "texture_loss"), (dl, "depth_loss"), (ddiff, "depth_diff"),
                     (ml, "match_loss"), (il, "imatch_loss")):
        np.testing.assert_allclose(got.detach().numpy(), d[key], rtol=2e-6, atol=1e-7, err_msg=key)
    (m.sum() + t.sum() + dl.sum() + ml.sum() + il.sum()).backward()
    for x, key in ((mask_pred, "g_mask_pred"), (tex_pred, "g_tex_pred"), (tex_mask, "g_tex_mask"),
                   (depth_pred, "g_depth_pred"), (match, "g_match"), (imatch, "g_imatch")):
        np.testing.assert_allclose(x.grad.numpy(), d[key], rtol=1e-5, atol=1e-8, err_msg=key)


def test_regularisers_and_pairing_match_reference():
    from scp_amd import losses as L
    d = golden_io.load("losses_small")
    lap = L.LaplacianLoss(_t(d["lap_verts"]), torch.tensor(d["lap_faces"]), average=True)
    assert abs(float(lap(_t(d["lap_in"]))) - float(d["lap_out"])) <= 1e-6 * abs(float(d["lap_out"]))
    x = _t(d["div_x"])
    sb, tb = L.divide_by_both(x, 2, 4)
    np.testing.assert_array_equal(sb.numpy(), d["div_both_src"])
    np.testing.assert_array_equal(tb.numpy(), d["div_both_tgt"])
    np.testing.assert_array_equal(L.divide_by_frame(x, 2, 4)[1].numpy(), d["div_frame_tgt"])
    np.testing.assert_array_equal(L.divide_by_instance(x, 2, 4)[1].numpy(), d["div_inst_tgt"])
    src_idx, tgt_idx = L.pair_indices("both", 2, 4)
    np.testing.assert_array_equal(x[src_idx].numpy(), d["div_both_src"])
    np.testing.assert_array_equal(x[tgt_idx].numpy(), d["div_both_tgt"])
    np.testing.assert_allclose(L.pinhole_cam(_t(d["cam_verts"]), _t(d["cam_pp"]), _t(d["cam_foc"])).numpy(), d["cam_out"], rtol=1e-6)


def test_eval_mode_confidence_matches_reference(monkeypatch):
    import oracle_backend
    oracle_backend.install(monkeypatch)
    from scp_amd.correspondence import Correspondence
    from scp_amd.flags import Options
    d = golden_io.load("corr_eval_conf_b2")
    corr = Correspondence(Options(corr_h=16, corr_w=16, n_corr_feat=16, train=False, tau_img=10., tau_mesh=10.))
    with torch.no_grad():
        _, match, imatch, conf = corr.match(_t(d["img_feat"]), _t(d["mesh_feat"]), _t(d["mask"]), _t(d["pred_v"]))
    np.testing.assert_allclose(match.numpy(), d["match"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(imatch.numpy(), d["imatch"], rtol=1e-5, atol=1e-6)
    # the nearest-vertex argmin can flip at exact near-ties of the L2 expansion; allow a handful of pixels
    bad = np.abs(conf.numpy() - d["match_conf"]) > 1e-4
    assert bad.mean() < 2e-3, bad.mean()


def test_flatten_loss_matches_reference():
    """loss_utils.py:98-171 incl. its edge enumeration (slots (0,1),(1,2) only, :105) -- off in the shipped configs
    (flatten_loss=False) but part of MeshNet's loss surface (model.py:116-119)"""
    from scp_amd import losses as L
    d = golden_io.load("flatten_loss_small")
    for tag in ("ico1", "bottle2"):
        fl = L.FlattenLoss(torch.tensor(d[tag + "_faces"]), average=True)
        assert fl.v0s.numel() == int(d[tag + "_n_edges"])
        x = _t(d[tag + "_verts"], True)
        y = fl(x)
        y.backward()
        assert abs(float(y) - float(d[tag + "_loss"])) <= 2e-6 * abs(float(d[tag + "_loss"]))
        # the edge ORDER differs (python set iteration in the reference), so index_add accumulates in another order
        np.testing.assert_allclose(x.grad.numpy(), d[tag + "_grad"], rtol=1e-4, atol=1e-6 * np.abs(d[tag + "_grad"]).max())
