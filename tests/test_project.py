"""The step's camera glue: loss_utils.py:38-61 (render(): bmm + translation, pinhole_cam, y flip) as the fused HIP projection
(csrc/project.hip, scp_amd.losses.project_vertices) and the two exact shortcuts of the SoftRenderer stages the trainer's cameras make
trivial (identity look-at rotation, unit lighting).  CPU: the shortcuts are bit-identical to the long way round.  GPU: the fused op vs the
torch composition in float64 -- forward to the rounding of the final cast (2 ulp), gradients 1e-6 relative; float64 and float32
intrinsics, both y conventions, detached operands, and a whole render pass with and without the fused glue."""
import math

import numpy as np
import pytest
import torch


def _camera(b, v, seed, f64):
    g = torch.Generator().manual_seed(seed)
    verts = torch.randn(b, v, 3, generator=g) * 0.4
    q = torch.randn(b, 3, 3, generator=g)
    rot = torch.linalg.qr(q).Q.contiguous()
    trans = torch.cat((torch.randn(b, 1, 2, generator=g) * 0.1, 4.5 + torch.rand(b, 1, 1, generator=g)), 2)
    dt = torch.float64 if f64 else torch.float32
    foc = (5.5 + torch.rand(b, 2, generator=g)).to(dt)
    pp = (torch.rand(b, 2, generator=g) * 0.3 - 0.15).to(dt)
    return verts, rot, trans, foc, pp


def test_look_at_and_lighting_shortcuts_are_bit_identical_cpu():
    from scp_amd.soft_renderer import functional as srf
    from scp_amd.soft_renderer.renderer import Lighting, LookAt
    from scp_amd.soft_renderer.mesh import Mesh
    la = LookAt(perspective=False)
    assert la._pure_translation()
    v = torch.randn(3, 50, 3) * 2
    long_way = srf.orthogonal(srf.look_at(v, la._eye), 1.0)
    assert torch.equal(la(v), long_way)
    assert not LookAt(perspective=True)._pure_translation()
    assert not LookAt(perspective=False, viewing_scale=2.0)._pure_translation()
    assert not LookAt(perspective=False, eye=[0.1, 0, -2.7])._pure_translation()
    other = LookAt(perspective=False, eye=[0.3, -0.2, -2.0])
    assert torch.allclose(other(v), srf.orthogonal(srf.look_at(v, other._eye), 1.0))
    faces = torch.randint(0, 50, (3, 20, 3))
    tex = torch.rand(3, 50, 3)
    lit = Lighting("vertex", 1., (1, 1, 1), 0.)
    assert lit._is_identity()
    m = lit(Mesh(v, faces, tex, texture_type="vertex"))
    assert torch.equal(m.textures, tex)
    dim = Lighting("vertex", 0.5, (1, 1, 1), 0.)
    assert not dim._is_identity() and torch.equal(dim(Mesh(v, faces, tex, texture_type="vertex")).textures, tex * 0.5)


def _reference(verts, rot, trans, foc, pp, flip_y):
    cam = verts.double().bmm(rot.double()) + trans.double()
    x = pp.double()[:, 0][:, None] + cam[:, :, 0] * foc.double()[:, 0][:, None] / cam[:, :, 2]
    y = pp.double()[:, 1][:, None] + cam[:, :, 1] * foc.double()[:, 1][:, None] / cam[:, :, 2]
    return torch.stack((x, -y if flip_y else y, cam[:, :, 2]), 2)


@pytest.mark.gpu
@pytest.mark.parametrize("b,v", [(32, 642), (3, 7), (2, 2562), (1, 257)])
@pytest.mark.parametrize("f64", [True, False])
@pytest.mark.parametrize("flip_y", [True, False])
def test_fused_projection_vs_float64_composition(b, v, f64, flip_y):
    from scp_amd import losses
    verts, rot, trans, foc, pp = (t.cuda() for t in _camera(b, v, b + v, f64))
    w = torch.randn(b, v, 3, generator=torch.Generator().manual_seed(1)).cuda()
    a = [t.clone().requires_grad_(True) for t in (verts, rot, trans)]
    out = losses.project_vertices(a[0], foc, pp, a[1], a[2], flip_y)
    assert out.grad_fn is not None and "ProjectVertices" in type(out.grad_fn).__name__
    (out * w).sum().backward()
    r = [t.double().clone().requires_grad_(True) for t in (verts, rot, trans)]
    ref = _reference(r[0], r[1], r[2], foc, pp, flip_y)
    (ref * w.double()).sum().backward()
    # forward: fp32 K = 3 product (a few ulp of the camera-space point) + the final rounding
    err = (out.double() - ref).abs()
    assert (err <= 2e-6 * (1 + ref.abs())).all(), err.max().item()
    for got, want, name in zip(a, r, ("verts", "rot", "trans")):
        scale = float(want.grad.abs().max())
        assert float((got.grad.double() - want.grad.reshape(got.grad.shape)).abs().max()) <= 2e-6 * scale, name
    # detached operands: only the requested gradients are produced
    a2 = verts.clone().requires_grad_(True)
    losses.project_vertices(a2, foc, pp, rot, trans, flip_y).sum().backward()
    assert a2.grad is not None


@pytest.mark.gpu
def test_fused_projection_equals_the_torch_composition_in_a_render_pass(monkeypatch):
    """one depth-group render with the fused glue and with FUSE_PROJECTION off: the projected vertices agree to the K = 3 summation
    order (1 ulp), the rendered images to the F12 band, image sums to 1e-4"""
    import scp_amd.dino as dino
    from scp_amd import losses, synthetic
    from scp_amd.flags import Options
    from scp_amd.model import MeshNet
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True)
    torch.manual_seed(0)
    model = MeshNet(opts, prior=synthetic.bottle_like(3)).cuda()
    data = synthetic.make_batch(1, 2, 256, seed=4, device="cuda")
    verts = model.mesh.mean_v.detach()[None].expand(2, -1, -1).contiguous() * 0.9
    _, rot, trans, _, _ = (t.cuda() for t in _camera(2, 4, 9, False))
    trans = torch.tensor([[[0.02, -0.03, 5.0]], [[-0.04, 0.01, 5.4]]], device="cuda")
    faces = model.mesh.faces[None].expand(2, -1, -1)
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(losses, "FUSE_PROJECTION", fused)
        with torch.no_grad():
            proj = losses.project_for_render(verts, data[7], data[9], rot, trans)
            depth_out, match_out, imatch_gt, dw = model.renderer.render_depth_group(verts, faces, data[7], data[9], rot, trans, raw=True)
        outs[fused] = (proj, depth_out, imatch_gt, dw)
    a, b = outs[True], outs[False]
    assert float((a[0] - b[0]).abs().max()) <= 1e-6
    assert float((a[2] - b[2]).abs().max()) <= 1e-6
    assert abs(float(a[1][:, 3].sum()) - float(b[1][:, 3].sum())) <= 1e-4 * float(b[1][:, 3].sum())      # F12: image sums 2e-6 .. 2.4e-5
    assert float(((a[1] - b[1]).abs() <= 1e-4).float().mean()) >= 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("poison", [None, "nan", "inf"])
def test_fused_gradient_clip_equals_the_torch_composition(poison):
    """trainer.py:132-150 on the flat buffer: csrc/gradclip.hip (two launches) vs the torch composition it replaces -- group norms,
    clipped values, untouched groups, the NaN guard; through Trainer.collect_grad on identical gradients"""
    import scp_amd.dino as dino
    from scp_amd import synthetic
    from scp_amd.flags import Options
    from scp_amd.trainer import Trainer
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True, total_iters=10)
    torch.manual_seed(0)
    tr = Trainer(opts, prior=synthetic.bottle_like(2), device="cuda")
    gen = torch.Generator().manual_seed(4)
    grads = [torch.randn(p.shape, generator=gen).cuda() * (3.0 if "pose" in n else 0.05)
             for n, p in tr.model.named_parameters() if p.requires_grad]
    res = {}
    for fused in (True, False):
        tr.fuse_clip = fused
        tr.grads.prepare()
        for p, g in zip(tr._trainable, grads):
            p.grad.copy_(g)
        if poison is not None:
            tr.grads.flat[tr.grads.flat.numel() // 3] = float(poison)
        norms = tr.collect_grad()
        res[fused] = (torch.stack([n.reshape(()) for n in norms]).cpu(), tr.grads.flat.detach().clone().cpu())
    assert hasattr(tr, "last_clip"), "the fused path must have run"
    np.testing.assert_allclose(res[True][0].numpy(), res[False][0].numpy(), rtol=2e-6, atol=0)
    a, b = res[True][1], res[False][1]
    assert torch.isfinite(a).all()
    if poison is not None:
        assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0 and float(res[True][0].abs().max()) == 0.0
    else:
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
        assert float(res[True][0][2]) > 0.1 and float(tr.last_clip[5]) < 1.0        # the pose group really was clipped


@pytest.mark.gpu
@pytest.mark.parametrize("offset,n", [(0, 100003), (1, 65537), (3, 4096 * 5 + 2), (2, 5), (0, 3)])
@pytest.mark.parametrize("poison", [False, True])
def test_gradclip_entry_point_on_unaligned_odd_buffers(offset, n, poison):
    """scp_gradclip through the C ABI on a buffer that starts off a 16-byte boundary, with a length that is no multiple of 4 and ranges
    that begin / end inside a float4 chunk (1- and 2-element ranges, a gap, adjacent ranges of different groups) -- the vector body, its
    head / tail lanes and the element-wise branch of a straddling chunk against float64 on the host; twice the same bits"""
    import ctypes
    from scp_amd import capi
    L = capi.lib()
    gen = torch.Generator().manual_seed(n + offset)
    base = torch.randn(n + 8, generator=gen).cuda()
    ranges = [(0, min(1, n), 0), (min(1, n), min(3, n), 1), (min(6, n), min(n, 4099), 2), (min(n, 4099), min(n, 4101), 0),
              (min(n, 50001), n, 1)]
    ranges = [r for r in ranges if r[1] > r[0]]
    mx = (0.05, 0.7, 1e9)
    ws_bytes = L.scp_gradclip_workspace()
    outs = []
    for rep in range(2):
        buf = base.clone()
        flat = buf[offset:offset + n]
        assert flat.data_ptr() % 16 == (4 * offset) % 16
        if poison:
            flat[n // 2] = float("nan")
        ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.float64, device="cuda")
        result = torch.zeros(8, device="cuda")
        arr = lambda vals, t: (t * len(vals))(*vals)
        capi.check(L.scp_gradclip(ctypes.c_void_p(flat.data_ptr()), n, ctypes.c_float(0.5), arr([r[0] for r in ranges], ctypes.c_longlong),
                                  arr([r[1] for r in ranges], ctypes.c_longlong), arr([r[2] for r in ranges], ctypes.c_int), len(ranges),
                                  ctypes.c_float(mx[0]), ctypes.c_float(mx[1]), ctypes.c_float(mx[2]), ctypes.c_void_p(ws.data_ptr()),
                                  ws_bytes, ctypes.c_void_p(result.data_ptr()), capi.current_stream()), "scp_gradclip")
        outs.append((buf.cpu(), result.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    got, res = outs[0]
    assert torch.equal(got[:offset], base.cpu()[:offset]) and torch.equal(got[offset + n:], base.cpu()[offset + n:])    # nothing outside
    if poison:
        assert float(got[offset:offset + n].abs().max()) == 0.0 and float(res[6]) == 0.0 and float(res[:3].abs().max()) == 0.0
        return
    g = base.cpu()[offset:offset + n].double() * 0.5
    ss = [0.0, 0.0, 0.0]
    grp = torch.full((n,), -1, dtype=torch.long)
    for b, e, k in ranges:
        ss[k] += float((g[b:e] ** 2).sum())
        grp[b:e] = k
    want = g.clone()
    for k in range(3):
        norm = np.float32(np.sqrt(ss[k]))
        coef = min(np.float32(mx[k]) / (norm + np.float32(1e-6)), np.float32(1.0))
        assert abs(float(res[k]) - float(norm)) <= 2e-6 * max(float(norm), 1e-30)
        assert abs(float(res[3 + k]) - float(coef)) <= 4e-6 * float(coef)
        want[grp == k] *= float(res[3 + k])
    assert float(res[6]) == 1.0
    assert float((got[offset:offset + n].double() - want).abs().max()) <= 2e-7 * max(float(want.abs().max()), 1e-30)


@pytest.mark.gpu
def test_flat_adamw_equals_torch_adamw():
    """optimizers.py:77-79: scp_amd.optimizers.FlatAdamW (one launch over FlatGradients' buffer, csrc/adamw.hip) against torch's
    fused AdamW on the same gradients for 6 steps -- channels_last convolution weights, odd sizes, per-group learning rates that
    change every step (OneCycle), a parameter that never gets a gradient and one that starts getting them late (own step count);
    parameters to 2e-6 of scale, and a state_dict round trip continues identically"""
    import copy
    import torch.nn as nn
    from scp_amd.optimizers import FlatAdamW
    from scp_amd.parallel import FlatGradients
    torch.manual_seed(0)

    def build():
        torch.manual_seed(1)
        net = nn.ModuleDict({"conv": nn.Conv2d(16, 33, 3), "lin": nn.Linear(77, 5), "late": nn.Linear(9, 9), "never": nn.Linear(4, 4),
                             "big": nn.Linear(300, 301)}).cuda()
        net["conv"].to(memory_format=torch.channels_last)
        return net

    a, b = build(), build()
    groups = lambda n: [{"params": list(n["conv"].parameters()) + list(n["late"].parameters())},
                        {"params": list(n["lin"].parameters()) + list(n["never"].parameters()) + list(n["big"].parameters()), "lr": 3e-3}]
    ref = torch.optim.AdamW(groups(a), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4, fused=True)
    own = FlatAdamW(groups(b), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)
    fg = FlatGradients(list(b.parameters()))
    assert own.attach(fg)
    sched = [torch.optim.lr_scheduler.OneCycleLR(o, [2e-3, 5e-3], total_steps=20, pct_start=0.3, cycle_momentum=False) for o in (ref, own)]
    gen = torch.Generator().manual_seed(7)

    def one_step(step, ref_opt, own_opt, scheds):
        fg.prepare()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            pa.grad = None
            if n.startswith("never") or (n.startswith("late") and step < 3):
                continue
            g = torch.randn(pa.shape, generator=gen).cuda()
            pa.grad = g.clone().contiguous(memory_format=torch.channels_last) if pa.dim() == 4 else g.clone()
            pb.grad.copy_(g)
        # what FlatGradients.finish(keep_unused_none=True) leaves behind: grad = None for parameters without a gradient this step
        for (n, _), pb in zip(a.named_parameters(), b.parameters()):
            if n.startswith("never") or (n.startswith("late") and step < 3):
                pb.grad = None
        ref_opt.step(); own_opt.step()
        for s_ in scheds:
            s_.step()

    for step in range(6):
        one_step(step, ref, own, sched)
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        scale = float(pa.abs().max())
        assert float((pa - pb).abs().max()) <= 2e-6 * scale, (n, float((pa - pb).abs().max()), scale)
    # the state travels: torch's layout out, our flat buffers back in
    sd = copy.deepcopy(own.state_dict())
    assert len(sd["state"]) == sum(1 for n, _ in b.named_parameters() if not n.startswith("never"))
    k0 = next(iter(sd["state"]))
    assert set(sd["state"][k0]) == {"step", "exp_avg", "exp_avg_sq"}
    own.load_state_dict(sd)
    ref.load_state_dict(copy.deepcopy(ref.state_dict()))
    for step in range(6, 9):
        one_step(step, ref, own, sched)
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert float((pa - pb).abs().max()) <= 4e-6 * float(pa.abs().max()), n


@pytest.mark.gpu
def test_flat_adamw_state_reset_and_load_then_attach():
    """(i) loading a state WITHOUT entries onto a stepped FlatAdamW resets steps and moments like torch's AdamW (which would
    find no state and start over); (ii) a state loaded BEFORE attach() is imported into the flat buffers, not dropped"""
    import copy
    import torch.nn as nn
    from scp_amd.optimizers import FlatAdamW
    from scp_amd.parallel import FlatGradients

    def build():
        torch.manual_seed(3)
        return nn.Linear(40, 7).cuda()

    gen = torch.Generator().manual_seed(11)
    grads = [[torch.randn(p.shape, generator=gen).cuda() for p in build().parameters()] for _ in range(6)]

    def run(opt, net, fg, steps):
        for k in steps:
            if fg is not None:
                fg.prepare()
            for p, g in zip(net.parameters(), grads[k]):
                if fg is not None:
                    p.grad.copy_(g)
                else:
                    p.grad = g.clone()
            opt.step()

    kw = dict(lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4)
    # (i) empty state after three steps == a fresh optimizer on the same parameters
    a, b = build(), build()
    ref, own = torch.optim.AdamW(a.parameters(), fused=True, **kw), FlatAdamW(b.parameters(), **kw)
    fg = FlatGradients(list(b.parameters()))
    assert own.attach(fg)
    run(ref, a, None, range(3)); run(own, b, fg, range(3))
    empty = {"state": {}, "param_groups": copy.deepcopy(own.state_dict()["param_groups"])}
    own.load_state_dict(empty)
    ref = torch.optim.AdamW(a.parameters(), fused=True, **kw)
    assert own._steps == [0, 0] and float(own._m.abs().max()) == 0.0 and float(own._v.abs().max()) == 0.0
    run(ref, a, None, range(3, 6)); run(own, b, fg, range(3, 6))
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert float((pa - pb).abs().max()) <= 4e-6 * float(pa.abs().max())
    # (ii) load first, attach afterwards
    sd = copy.deepcopy(own.state_dict())
    c = build()
    with torch.no_grad():
        for pc, pb in zip(c.parameters(), b.parameters()):
            pc.copy_(pb)
    late = FlatAdamW(c.parameters(), **kw)
    late.load_state_dict(sd)
    fg2 = FlatGradients(list(c.parameters()))
    assert late.attach(fg2)
    assert late._steps == own._steps and torch.equal(late._m, own._m) and torch.equal(late._v, own._v)
    run(own, b, fg, [0]); run(late, c, fg2, [0])
    for pb, pc in zip(b.parameters(), c.parameters()):
        assert torch.equal(pb, pc)
