"""Dense correspondence operators.

CPU (not gpu): the oracle (oracle/corr.py) and the product's host definitions (scp_amd.ops /
Correspondence) reproduce the golden vectors recorded from the reference's Correspondence class.
GPU (-m gpu): the HIP kernels (csrc/corr.hip through the C ABI + autograd nodes) agree with the oracle
on the golden inputs, on ragged shapes, with fully masked rows/columns, and at the full bench size.

Tolerance: these are fp32 softmax reductions; summation order differs from torch's (wavefront trees,
chunked online softmax), so |d| <= 2e-6 + 1e-5|ref| forward and 1e-4 of the gradient scale backward
(north_star: 1e-4 relative)."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import corr as oracle


def _t(a, dev="cpu", grad=False):
    return torch.tensor(a, device=dev, requires_grad=grad)


def _close(got, ref, rtol=1e-5, atol=2e-6):
    got, ref = got.detach().cpu().double().numpy(), np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    assert (d <= atol + rtol * np.abs(ref)).all(), "max abs diff %.3e (ref scale %.3e)" % (d.max(), np.abs(ref).max())


def _grad_close(got, ref, tol=1e-4):
    got, ref = got.detach().cpu().double().numpy(), np.asarray(ref, np.float64)
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), "max abs diff %.3e vs scale %.3e" % (
        np.abs(got - ref).max(), np.abs(ref).max())


def _up(match, hf=16, size=64):
    """correspondence.py:71: nearest upsampling of the per-pixel match to image resolution"""
    b = match.shape[0]
    return torch.nn.functional.interpolate(match.reshape(b, hf, hf, 3).permute(0, 3, 1, 2), (size, size), mode="nearest")


def _match_case(dev):
    d = golden_io.load("corr_match_b2")
    img_feat, mesh_feat = _t(d["img_feat"], dev, True), _t(d["mesh_feat"], dev, True)
    mask_down = torch.nn.functional.interpolate(_t(d["mask"], dev)[:, None], (16, 16), mode="nearest").reshape(2, -1)
    return d, img_feat, mesh_feat, mask_down


def test_oracle_match_reproduces_reference():
    d, img_feat, mesh_feat, mask_down = _match_case("cpu")
    pc, match, imatch = oracle.match_oracle(img_feat, mesh_feat, mask_down, _t(d["pred_v"]), _t(d["meshgrid"]), 10., 10.)
    np.testing.assert_array_equal(pc.detach().numpy(), d["pointcorr"])
    match = _up(match)
    np.testing.assert_array_equal(match.detach().numpy(), d["match"])
    np.testing.assert_array_equal(imatch.detach().numpy(), d["imatch"])
    ((pc * _t(d["w_pc"])).sum() * 1e-2 + (match * _t(d["w_match"])).sum() + (imatch * _t(d["w_imatch"])).sum()).backward()
    np.testing.assert_allclose(img_feat.grad.numpy(), d["grad_img_feat"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mesh_feat.grad.numpy(), d["grad_mesh_feat"], rtol=1e-5, atol=1e-6)


def _run_product_match(dev):
    from scp_amd import ops
    d, img_feat, mesh_feat, mask_down = _match_case(dev)
    pc, match, imatch = ops.feature_vertex_match(img_feat, mesh_feat, mask_down, _t(d["pred_v"], dev),
                                                 _t(d["meshgrid"], dev), 10., 10.)
    match = _up(match)
    _close(pc, d["pointcorr"])
    _close(match, d["match"])
    _close(imatch, d["imatch"])
    ((pc * _t(d["w_pc"], dev)).sum() * 1e-2 + (match * _t(d["w_match"], dev)).sum() + (imatch * _t(d["w_imatch"], dev)).sum()).backward()
    _grad_close(img_feat.grad, d["grad_img_feat"])
    _grad_close(mesh_feat.grad, d["grad_mesh_feat"])


def test_host_match_reproduces_reference(monkeypatch):
    import oracle_backend
    oracle_backend.install(monkeypatch)
    _run_product_match("cpu")


def _run_rotation_cycle(dev, monkeypatch_rotate=None):
    from scp_amd.correspondence import Correspondence
    from scp_amd.flags import Options
    d = golden_io.load("corr_rotcycle_b2")
    opts = Options(corr_h=16, corr_w=16, n_corr_feat=16, train=True, tau_img=10., tau_mesh=10.)
    corr = Correspondence(opts, device=dev)
    src_feat, tgt_feat = _t(d["src_feat"], dev, True), _t(d["tgt_feat"], dev, True)

    class Enc:
        def encode_img(self, img):
            # the real encoder L2-normalises its output (encoder.py:36); the reference normalises once
            # more inside the loss (correspondence.py:93), which is idempotent in value and gradient
            return None, torch.nn.functional.normalize(tgt_feat, 2, 1)

    loss, cycle_match, cycle_match_gt, tgt_mask_down = corr.compute_rotation_cycle_loss(
        _t(d["src_img"], dev), _t(d["mask"], dev), src_feat, Enc(), angle=float(d["angle"]))
    np.testing.assert_array_equal(cycle_match_gt.cpu().numpy(), d["cycle_match_gt"])
    np.testing.assert_array_equal(tgt_mask_down.cpu().numpy(), d["tgt_mask_down"])
    _close(cycle_match, d["cycle_match"])
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * abs(float(d["loss"]))
    loss.backward()
    _grad_close(src_feat.grad, d["grad_src_feat"])
    _grad_close(tgt_feat.grad, d["grad_tgt_feat"])


def test_host_rotation_cycle_reproduces_reference(monkeypatch):
    import oracle_backend
    oracle_backend.install(monkeypatch)
    _run_rotation_cycle("cpu")


def test_reassociated_bridge_equals_reference_formulation(monkeypatch):
    """ops.vertex_bridge_match (no [N,P,P] matrix) == the reference's corr-matrix formulation"""
    import oracle_backend
    oracle_backend.install(monkeypatch)
    from scp_amd import ops
    g = torch.Generator().manual_seed(5)
    B, hf, V, K = 4, 8, 37, 9
    pc = torch.randn(B, hf * hf, V, generator=g).requires_grad_(True)
    dw = torch.rand(B, V, generator=g)
    src_idx, tgt_idx = torch.tensor([0, 1, 2, 3, 0, 2]), torch.tensor([1, 0, 3, 2, 2, 0])
    n = src_idx.shape[0]
    grid_half = torch.rand(2, (hf // 2) ** 2, generator=g) * 2 - 1
    idx_t = torch.stack([torch.randperm((hf // 2) ** 2, generator=g)[:K] for _ in range(n)])
    pts_src, mask = torch.rand(n, 2, K, generator=g), (torch.rand(n, K, generator=g) > 0.2).float()
    ref_loss, ref_match = oracle.bridge_cycle_oracle(pc[src_idx], pc[tgt_idx], dw[src_idx], dw[tgt_idx],
                                                     grid_half[None].expand(n, -1, -1), idx_t, pts_src, mask, hf, hf, 10., 10.)
    g_ref, = torch.autograd.grad(ref_loss, pc)
    pooled = ops.pool2x2_scores(pc, hf, hf)
    got = ops.vertex_bridge_match(pooled, src_idx, tgt_idx, idx_t, dw >= 0.5, grid_half, 10., 10.)
    got_loss = ((got - pts_src).norm(2, 1) * mask).mean()
    g_got, = torch.autograd.grad(got_loss, pc)
    _close(got, ref_match.detach().numpy(), rtol=1e-5, atol=1e-6)
    _grad_close(g_got, g_ref.numpy(), 1e-4)


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_match_reproduces_reference():
    _run_product_match("cuda")


@pytest.mark.gpu
def test_hip_rotation_cycle_reproduces_reference():
    _run_rotation_cycle("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,P,V", [(1, 8, 70, 5), (3, 64, 1024, 642), (2, 16, 300, 1100)])
def test_hip_match_vs_oracle_ragged(B, C, P, V):
    """odd sizes (P, V not multiples of 64; V > the 1024-column register cache), images with an
    all-zero mask (every score masked: uniform softmax over -1e5) and with an all-one mask"""
    from scp_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + V)
    img = torch.nn.functional.normalize(torch.randn(B, C, P, generator=g), 2, 1)
    mesh = torch.nn.functional.normalize(torch.randn(B, V, C, generator=g), 2, 2)
    verts, grid = torch.randn(B, V, 3, generator=g), torch.rand(2, P, generator=g) * 2 - 1
    mask = (torch.rand(B, P, generator=g) > 0.4).float()
    mask[0] = 0.
    if B > 1:
        mask[1] = 1.
    w_pc, w_m, w_i = torch.randn(B, P, V, generator=g), torch.randn(B, P, 3, generator=g), torch.randn(B, 2, V, generator=g)

    def run(dev, fn):
        a, b = img.detach().clone().to(dev).requires_grad_(True), mesh.detach().clone().to(dev).requires_grad_(True)
        pc, match, imatch = fn(a, b, mask.to(dev), verts.to(dev), grid.to(dev), 10., 10.)
        live = (pc.detach() > -1e4).float()
        ((pc * w_pc.to(dev) * live).sum() * 1e-2 + (match * w_m.to(dev)).sum() + (imatch * w_i.to(dev)).sum()).backward()
        return pc, match, imatch, a.grad, b.grad

    ref = run("cpu", oracle.match_oracle)
    got = run("cuda", ops.feature_vertex_match)
    for r, h in zip(ref[:3], got[:3]):
        _close(h, r.detach().numpy())
    _grad_close(got[3], ref[3].numpy())
    _grad_close(got[4], ref[4].numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("N,P,Q,batched,masks", [(2, 1024, 1024, True, True), (3, 64, 96, False, True), (1, 160, 32, True, False)])
def test_fused_pixel_pixel_softargmax_vs_oracle(N, P, Q, batched, masks):
    """csrc/corr_pp.hip (a10: scores never stored) against the materialised oracle formulation of correspondence.py:105-110 and
    against the build's own unfused path (library GEMM + csrc/corr.hip): output and both feature gradients; source and target
    masks, one image with an empty source mask (uniform softmax), P != Q, fewer source tiles than wavefronts, shared grid"""
    from scp_amd import corr_ops, ops
    g = torch.Generator().manual_seed(N * 1000 + P + Q)
    src = torch.nn.functional.normalize(torch.randn(N, 64, P, generator=g), 2, 1)
    tgt = torch.nn.functional.normalize(torch.randn(N, 64, Q, generator=g), 2, 1)
    sm = (torch.rand(N, P, generator=g) > 0.4).float() if masks else None
    tm = (torch.rand(N, Q, generator=g) > 0.4).float() if masks else None
    if masks:
        sm[0] = 0.
    grid = torch.rand(*((N, 2, P) if batched else (2, P)), generator=g) * 2 - 1
    w = torch.randn(N, 2, Q, generator=g)
    assert corr_ops.pp_fusable(src.cuda(), tgt.cuda())

    def run_ref():
        a, b = src.clone().double().requires_grad_(True), tgt.clone().double().requires_grad_(True)
        out = oracle.cols_softargmax_oracle(a.transpose(1, 2).bmm(b), sm, tm, grid.double(), 10.)
        (out * w.double()).sum().backward()
        return out, a.grad, b.grad

    def run_hip(fused):
        a, b = src.clone().cuda().requires_grad_(True), tgt.clone().cuda().requires_grad_(True)
        m = (None if sm is None else sm.cuda(), None if tm is None else tm.cuda())
        if fused:
            out = ops.pixel_pixel_softargmax(a, b, m[0], m[1], grid.cuda(), 10.)
        else:
            out = ops.cols_softargmax(a.transpose(1, 2).bmm(b), m[0], m[1], grid.cuda(), 10.)
        (out * w.cuda()).sum().backward()
        return out, a.grad, b.grad

    ref, got, two_step = run_ref(), run_hip(True), run_hip(False)
    _close(got[0], ref[0].detach().numpy())
    _grad_close(got[1], ref[1].numpy())
    _grad_close(got[2], ref[2].numpy())
    # not further from float64 than the path it replaces (up to a factor for the different summation order)
    for x, y, r in zip(got, two_step, ref):
        r = r.detach().numpy()
        e_new = np.abs(x.detach().cpu().double().numpy() - r).max()
        e_old = np.abs(y.detach().cpu().double().numpy() - r).max()
        assert e_new <= 4 * e_old + 1e-7 * np.abs(r).max(), (e_new, e_old)


@pytest.mark.gpu
def test_hip_cols_softargmax_masks_and_batched_grid():
    from scp_amd import ops
    g = torch.Generator().manual_seed(3)
    N, P, Q = 3, 200, 130
    s = torch.randn(N, P, Q, generator=g)
    rm, cm = (torch.rand(N, P, generator=g) > 0.3).float(), (torch.rand(N, Q, generator=g) > 0.3).float()
    grid = torch.rand(N, 2, P, generator=g)
    w = torch.randn(N, 2, Q, generator=g)

    def run(dev):
        x = s.detach().clone().to(dev).requires_grad_(True)
        out = ops.cols_softargmax(x, rm.to(dev), cm.to(dev), grid.to(dev), 10.)
        (out * w.to(dev)).sum().backward()
        return out, x.grad

    def run_ref():
        x = s.detach().clone().requires_grad_(True)
        out = oracle.cols_softargmax_oracle(x, rm, cm, grid, 10.)
        (out * w).sum().backward()
        return out, x.grad

    (o_ref, g_ref), (o_hip, g_hip) = run_ref(), run("cuda")
    _close(o_hip, o_ref.detach().numpy())
    _grad_close(g_hip, g_ref.numpy())


@pytest.mark.gpu
def test_hip_match_full_size_properties():
    """BASELINE size B=32, P=4096, V=642, C=64: soft-argmax outputs are convex combinations
    (inside the hull of their inputs), permutation-equivariant over the batch, deterministic, and
    agree with the oracle on a 2-image subset"""
    from scp_amd import ops
    g = torch.Generator().manual_seed(0)
    B, C, P, V = 32, 64, 4096, 642
    img = torch.nn.functional.normalize(torch.randn(B, C, P, generator=g), 2, 1).cuda()
    mesh = torch.nn.functional.normalize(torch.randn(B, V, C, generator=g), 2, 2).cuda()
    verts = torch.randn(B, V, 3, generator=g).cuda()
    xs = (torch.arange(64.) + 0.5) / 32 - 1
    grid = torch.stack((xs.repeat(64), xs.repeat_interleave(64))).cuda()
    mask = (torch.rand(B, P, generator=g) > 0.5).float().cuda()
    pc, match, imatch = ops.feature_vertex_match(img, mesh, mask, verts, grid, 10., 10.)
    assert imatch.abs().max() <= 1.0 and torch.isfinite(match).all()
    assert (match.amax(1) <= verts.amax(1) + 1e-5).all() and (match.amin(1) >= verts.amin(1) - 1e-5).all()
    perm = torch.randperm(B, generator=g).cuda()
    pc2, match2, imatch2 = ops.feature_vertex_match(img[perm], mesh[perm], mask[perm], verts[perm], grid, 10., 10.)
    assert torch.equal(match2, match[perm]) and torch.equal(imatch2, imatch[perm]) and torch.equal(pc2, pc[perm])
    sub = [3, 17]
    r_pc, r_match, r_imatch = oracle.match_oracle(img[sub].cpu(), mesh[sub].cpu(), mask[sub].cpu(), verts[sub].cpu(),
                                                  grid.cpu(), 10., 10.)
    _close(pc[sub], r_pc.numpy())
    _close(match[sub], r_match.numpy())
    _close(imatch[sub], r_imatch.numpy())


@pytest.mark.gpu
def test_hip_nearest_point_matches_bruteforce():
    """symmetry-loss 1-NN (csrc/nearest.hip) vs an exact torch brute force, incl. exact ties"""
    from scp_amd.mesh import nearest_sq_dist
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(5, 300, 3, generator=g), torch.randn(5, 2500, 3, generator=g)
    y[:, 100] = y[:, 50]                      # duplicate candidate: lowest index must win
    x[:, 0] = y[:, 50]
    d = (x[:, :, None] - y[:, None]).pow(2).sum(-1)
    ref_d, ref_i = d.min(-1)
    xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    got = nearest_sq_dist(xg, yg)
    torch.testing.assert_close(got.cpu(), ref_d, rtol=1e-6, atol=1e-7)
    from scp_amd.mesh import _nearest_index_hip
    assert torch.equal(_nearest_index_hip(x.cuda(), y.cuda()).cpu(), ref_i)
    got.sum().backward()
    assert torch.isfinite(xg.grad).all() and yg.grad.abs().sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,P,Q", [(3, 16, 64, 64), (2, 8, 37, 20), (2, 384, 1024, 1024), (1, 4, 5, 2052)])
def test_hip_mutual_argmax_vs_oracle(N, C, P, Q):
    """one-pass row/column argmax of the masked score matrix vs pretrained_corr.py:85-89 restated (oracle/corr.py),
    including exact ties (duplicated feature columns), fully masked rows/columns and a ragged row count"""
    from scp_amd import corr_ops
    g = torch.Generator().manual_seed(N * 1000 + P)
    src = torch.nn.functional.normalize(torch.randn(N, C, P, generator=g), dim=1)
    tgt = torch.nn.functional.normalize(torch.randn(N, C, Q, generator=g), dim=1)
    src[:, :, 3] = src[:, :, 1]                      # ties along rows
    tgt[:, :, min(7, Q - 1)] = tgt[:, :, 0]          # ties along columns
    sm = (torch.rand(N, P, generator=g) > 0.3).float()
    tm = (torch.rand(N, Q, generator=g) > 0.3).float()
    sm[0, :] = 0 if N > 2 else sm[0, :]              # a pair with nothing to match
    pc = src.transpose(1, 2).bmm(tgt)
    bw_ref, fw_ref, masked = oracle.mutual_nn_oracle(src, tgt, sm, tm)
    bw, fw = corr_ops.mutual_argmax(pc.cuda(), sm.cuda(), tm.cuda())
    bw, fw = bw.cpu(), fw.cpu()
    # same GEMM result on both sides (computed once on the CPU), so indices must agree exactly
    assert torch.equal(bw, bw_ref) and torch.equal(fw, fw_ref)
    bw2, fw2 = corr_ops.mutual_argmax(pc.cuda(), None, None)
    assert torch.equal(bw2.cpu(), pc.max(1).indices) and torch.equal(fw2.cpu(), pc.max(2).indices)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "fp32"])
@pytest.mark.parametrize("B,n_tok,C,tok0", [(4, 1025, 384, 1), (3, 201, 64, 1), (2, 4097, 384, 1), (2, 128, 32, 0)])
def test_hip_mutual_nn_fused_vs_float64(B, n_tok, C, tok0, mode):
    """scp_mutual_nn_fused (score GEMM + dual argmax in one kernel, no score tensor) against pretrained_corr.py:85-89 evaluated in
    float64 on the same keys: every selected index is THE argmax, or differs from it only at a near-tie of the float64 scores
    (top-2 gap < 1e-5 relative: SURVEY F16); exact ties (bitwise duplicated tokens) resolve to the lowest index in both
    directions; an image that is fully masked, pairs that repeat an image, P not a multiple of the 128-wide tile."""
    from scp_amd import corr_ops
    g = torch.Generator().manual_seed(B * 7 + n_tok)
    P = n_tok - tok0
    keys = torch.randn(B, n_tok, C, generator=g) * 1.7
    keys[:, tok0 + 3] = keys[:, tok0 + 1]                    # exact ties: token 3 duplicates token 1 in every image
    keys[:, tok0 + P - 1] = keys[:, tok0 + 2]
    mask = (torch.rand(B, P, generator=g) > 0.35).float()
    mask[:, 1] = mask[:, 3] = mask[:, 2] = mask[:, P - 1] = 1.   # both members of every duplicated pair take part
    mask[B - 1] = 0.                                         # an image with nothing to match
    src = torch.tensor([0, 1, 0, B - 1, 1][:max(2, B + 1)])
    tgt = torch.tensor([1, 0, 0, 0, B - 1][:max(2, B + 1)])
    bw, fw = corr_ops.mutual_nn_fused(keys.cuda(), src, tgt, mask.cuda(), tok0, mode)
    bw, fw = bw.cpu(), fw.cpu()
    k64 = keys[:, tok0:].double()
    for n in range(src.numel()):
        s, t = int(src[n]), int(tgt[n])
        pc = k64[s] @ k64[t].t()
        keep = (mask[s][:, None] > 0) & (mask[t][None, :] > 0)
        pc = torch.where(keep, pc, torch.full_like(pc, -1e5))
        for name, got, ref_v, ref_i, dim in (("bw", bw[n], pc.max(0).values, pc.max(0).indices, 0), ("fw", fw[n], pc.max(1).values, pc.max(1).indices, 1)):
            chosen = pc.gather(dim, got[None] if dim == 0 else got[:, None]).reshape(-1)
            gap = ref_v - chosen
            assert bool((gap >= 0).all())
            flips = got != ref_i
            assert bool((gap[flips] <= 1e-5 * ref_v[flips].abs()).all()), (name, n, gap[flips].max())
            assert flips.float().mean() <= 0.02, (name, n, flips.float().mean())
            # exact ties resolve to the lowest index: the duplicates (3 of 1, P-1 of 2) are never selected
            assert not bool((got == 3).any()) and not bool((got == P - 1).any()), (name, n)
        if mask[s].sum() == 0 or mask[t].sum() == 0:         # all entries -1e5: lowest index everywhere
            assert bool((bw[n] == 0).all()) and bool((fw[n] == 0).all())
    # no mask: plain argmax
    bw2, fw2 = corr_ops.mutual_nn_fused(keys.cuda(), src[:1], tgt[:1], None, tok0, mode)
    pc = k64[int(src[0])] @ k64[int(tgt[0])].t()
    assert (bw2[0].cpu() != pc.max(0).indices).float().mean() <= 0.02 and (fw2[0].cpu() != pc.max(1).indices).float().mean() <= 0.02


@pytest.mark.gpu
def test_hip_mutual_nn_fused_equals_two_step_path():
    """the fused kernel against the build's own two-step path (score tensor by a library GEMM of the same keys, then
    scp_mutual_argmax) at the training shape: identical wherever the two GEMMs' scores do not near-tie"""
    from scp_amd import corr_ops
    g = torch.Generator().manual_seed(3)
    B, n_tok, C = 8, 1025, 384
    keys = (torch.randn(B, n_tok, C, generator=g) * 1.5).cuda()
    mask = (torch.rand(B, 1024, generator=g) > 0.4).float().cuda()
    src = torch.arange(B).repeat(2)
    tgt = torch.cat((torch.arange(B).roll(1), torch.arange(B).roll(3)))
    bw, fw = corr_ops.mutual_nn_fused(keys, src, tgt, mask, 1, "split")
    feat = keys[:, 1:].transpose(1, 2)
    pc = feat[src].transpose(1, 2).bmm(feat[tgt])
    bw_ref, fw_ref = corr_ops.mutual_argmax(pc, mask[src], mask[tgt])
    assert (bw != bw_ref).float().mean() <= 1e-3 and (fw != fw_ref).float().mean() <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,hf,V", [(2, 64, 642), (3, 8, 37), (1, 16, 995), (2, 4, 64)])
def test_fused_feature_vertex_match_vs_oracle(B, hf, V):
    """csrc/corr_fused.hip (scores never stored) against the materialised oracle: pooled scores, match, imatch and the
    gradients w.r.t. both feature sets under random upstream gradients on all three outputs; V not a multiple of 32,
    one image fully masked, one fully visible"""
    from scp_amd import ops
    wf, C = 64, 64
    P = hf * wf
    g = torch.Generator().manual_seed(B * 100 + V)
    img = torch.nn.functional.normalize(torch.randn(B, C, P, generator=g), 2, 1)
    mesh = torch.nn.functional.normalize(torch.randn(B, V, C, generator=g), 2, 2)
    verts = torch.randn(B, V, 3, generator=g)
    xs, ys = (torch.arange(float(wf)) + 0.5) / (wf / 2) - 1, (torch.arange(float(hf)) + 0.5) / (wf / 2) - 1
    grid = torch.stack((xs.repeat(hf), ys.repeat_interleave(wf)))
    mask = (torch.rand(B, P, generator=g) > 0.4).float()
    mask[0] = 0.
    if B > 1:
        mask[1] = 1.
    w_p, w_m, w_i = torch.randn(B, P // 4, V, generator=g), torch.randn(B, P, 3, generator=g), torch.randn(B, 2, V, generator=g)

    def run_ref():
        a, b = img.clone().requires_grad_(True), mesh.clone().requires_grad_(True)
        pc, match, imatch = oracle.match_oracle(a, b, mask, verts, grid, 10., 10.)
        pooled = ops.pool2x2_scores(pc, hf, wf)
        live = (pooled.detach() > -1e4).float()           # cells with a masked pixel hold -2.5e4 ...: compare finite cells
        ((pooled * w_p * live).sum() * 1e-2 + (match * w_m).sum() + (imatch * w_i).sum()).backward()
        return pooled, match, imatch, a.grad, b.grad

    def run_hip():
        a, b = img.clone().cuda().requires_grad_(True), mesh.clone().cuda().requires_grad_(True)
        out = ops.feature_vertex_match_pooled(a, b, mask.cuda(), verts.cuda(), grid.cuda(), 10., 10., hf, wf)
        assert out is not None
        pc, match, imatch = out
        pooled = ops.pool2x2_scores(pc, hf, wf)
        assert pooled.shape == (B, P // 4, V)
        live = (pooled.detach() > -1e4).float()
        ((pooled * w_p.cuda() * live).sum() * 1e-2 + (match * w_m.cuda()).sum() + (imatch * w_i.cuda()).sum()).backward()
        return pooled, match, imatch, a.grad, b.grad

    ref, got = run_ref(), run_hip()
    # pooled: cells containing masked pixels are sums with -1e5 terms (absolute error ~1e-2 at that magnitude): relative
    np.testing.assert_allclose(got[0].detach().cpu().numpy(), ref[0].detach().numpy(), rtol=2e-6, atol=2e-6)
    _close(got[1], ref[1].detach().numpy())
    _close(got[2], ref[2].detach().numpy())
    _grad_close(got[3], ref[3].numpy())
    _grad_close(got[4], ref[4].numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("B,hf,V", [(3, 64, 642), (2, 8, 37)])
def test_fused_bridge_columns_equal_the_separate_pass(B, hf, V):
    """feature_vertex_match_pooled(grid_half=...) also leaves the column soft-argmax of the POOLED scores (the mesh -> image half of
    the vertex bridge, pretrained_corr.py:123-126): same values as the separate pass over pooled[], and vertex_bridge_match gives the
    same output and the same feature gradients whether it uses them or recomputes them; one fully masked image"""
    from scp_amd import ops
    wf, C, K = 64, 64, 9
    P = hf * wf
    g = torch.Generator().manual_seed(B * 7 + V)
    img = torch.nn.functional.normalize(torch.randn(B, C, P, generator=g), 2, 1).cuda()
    mesh = torch.nn.functional.normalize(torch.randn(B, V, C, generator=g), 2, 2).cuda()
    verts = torch.randn(B, V, 3, generator=g).cuda()
    xs, ys = (torch.arange(float(wf)) + 0.5) / (wf / 2) - 1, (torch.arange(float(hf)) + 0.5) / (wf / 2) - 1
    grid = torch.stack((xs.repeat(hf), ys.repeat_interleave(wf))).cuda()
    gh = torch.nn.functional.interpolate(grid.reshape(2, hf, wf)[None], (hf // 2, wf // 2), mode="bilinear").reshape(2, -1)
    mask = (torch.rand(B, P, generator=g) > 0.4).float().cuda()
    mask[0] = 0.
    src_idx, tgt_idx = torch.arange(B).cuda(), torch.arange(B).roll(1).cuda()
    pix = torch.stack([torch.randperm(P // 4, generator=g)[:K] for _ in range(B)]).cuda()
    keep = (torch.rand(B, V, generator=g) > 0.3).cuda()
    w = torch.randn(B, 2, K, generator=g).cuda()

    def run(with_bridge):
        a, b = img.clone().requires_grad_(True), mesh.clone().requires_grad_(True)
        pc, match, imatch = ops.feature_vertex_match_pooled(a, b, mask, verts, grid, 10., 10., hf, wf, gh if with_bridge else None)
        assert (pc.bridge is not None) == with_bridge
        out = ops.vertex_bridge_match(ops.pool2x2_scores(pc, hf, wf), src_idx, tgt_idx, pix, keep, gh, 10., 10., precomputed=pc.bridge)
        ((out * w).sum() + match.square().sum() * 1e-3).backward()
        return pc, out.detach(), a.grad, b.grad

    pc, out1, ga1, gb1 = run(True)
    ref = ops.cols_softargmax(pc.pooled.detach(), None, None, gh, 10.)
    _close(pc.bridge[2], ref.cpu().numpy())
    _, out0, ga0, gb0 = run(False)
    _close(out1, out0.cpu().numpy())
    _grad_close(ga1, ga0.cpu().numpy(), 2e-5)
    _grad_close(gb1, gb0.cpu().numpy(), 2e-5)


@pytest.mark.gpu
def test_fused_match_in_training_step_equals_unfused():
    """Correspondence.match (train mode) through the fused kernels vs the round-1 path (rocBLAS scores + reductions): same
    match / imatch, same pooled scores, same feature gradients"""
    from scp_amd import ops
    from scp_amd.correspondence import Correspondence
    from scp_amd.flags import Options
    opts = Options("laptop_wild6d", train=True)
    g = torch.Generator().manual_seed(5)
    B, V = 4, 642
    img = torch.nn.functional.normalize(torch.randn(B, 64, 4096, generator=g), 2, 1).cuda()
    mesh = torch.nn.functional.normalize(torch.randn(B, V, 64, generator=g), 2, 2).cuda()
    pred_v = torch.randn(B, V, 3, generator=g).cuda()
    mask = (torch.rand(B, 256, 256, generator=g) > 0.3).float().cuda()
    outs = []
    for fuse in (True, False):
        corr = Correspondence(opts, "cuda")
        corr.fuse_scores = fuse
        a, b = img.clone().requires_grad_(True), mesh.clone().requires_grad_(True)
        pc, match, imatch, _ = corr.match(a, b, mask, pred_v)
        pooled = ops.pool2x2_scores(pc, 64, 64)
        live = (pooled.detach() > -1e4).float()
        ((pooled * live).sum() * 1e-3 + match.square().sum() + imatch.square().sum()).backward()
        outs.append((pooled.detach(), match.detach(), imatch.detach(), a.grad, b.grad))
    for x, y in zip(*outs):
        scale = y.abs().max().item()
        assert (x - y).abs().max().item() <= 2e-5 * scale, ((x - y).abs().max().item(), scale)
