"""tests/coresidency.py -- engine of the co-residency screen (tests/test_coresidency_gpu.py).

On gfx950 packed-fp32 instructions with op_sel [0,1] compute wrong low halves while a K-doubled 16-bit MFMA runs on the same SIMD
(csrc/selftest.hip).  The screen puts a persistent register-only loop of v_mfma_f32_32x32x16_bf16 on a side stream -- one that leaves most of
every SIMD free, so the victim's wavefronts sit right beside it -- and runs a VICTIM (any callable returning tensors) many times on the main
stream with identical inputs.  A victim is clean when every pass reproduces its unloaded result: bit for bit where the unloaded passes are
bit-identical among themselves, else within a small multiple of the unloaded run-to-run spread (float atomics).

`python tests/coresidency.py [passes]` prints the table for every victim (profiles/r05_coresidency_screen.txt is its output)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from scp_amd import capi  # noqa: E402


class MfmaLoad:
    """with MfmaLoad(kind): ...   -- `blocks` workgroups looping one matrix instruction on a side stream for the duration of the block.
    kind 0 = v_mfma_f32_32x32x16_bf16 (the aggressor), 1 = v_mfma_f32_32x32x2_f32 (the control that must change nothing)."""

    _hip = None

    def __init__(self, kind=0, blocks=512, device="cuda"):
        self.kind, self.blocks = kind, blocks
        self.stream = torch.cuda.Stream(device=device)
        self.sink = torch.empty(blocks * 256, device=device)
        # the stop flag lives in COHERENT pinned host memory and is raised by a plain host store: a flag in device memory needs a kernel
        # on a third stream to raise it, and when that stream shares a hardware queue with the load stream the store waits behind every
        # queued load launch (round 6: four stage screens took 45 s each that way)
        if MfmaLoad._hip is None:
            MfmaLoad._hip = ctypes.CDLL("libamdhip64.so")
        ptr = ctypes.c_void_p()
        code = MfmaLoad._hip.hipHostMalloc(ctypes.byref(ptr), ctypes.c_size_t(64), ctypes.c_uint(0x40000000))    # hipHostMallocCoherent
        if code != 0 or not ptr.value:
            raise RuntimeError("hipHostMalloc(coherent) failed: %d" % code)
        self._stop_ptr = ptr.value
        self._stop = ctypes.c_int.from_address(ptr.value)
        self._stop.value = 0

    def __del__(self):
        try:
            if getattr(self, "_stop_ptr", None):
                MfmaLoad._hip.hipHostFree(ctypes.c_void_p(self._stop_ptr))
                self._stop_ptr = None
        except Exception:                               # noqa: BLE001 -- interpreter shutdown
            pass

    def __enter__(self):
        main = torch.cuda.current_stream()
        self._stop.value = 0
        self.stream.wait_stream(main)
        # 2^24 instructions bound one launch to ~0.3 s even if nobody raises the flag; re-armed by keep_alive()
        self._launch()
        return self

    def _launch(self):
        capi.check(capi.lib().scp_selftest_mfma_load(self.kind, ctypes.c_void_p(self.sink.data_ptr()), self.blocks, 1 << 24,
                                                    ctypes.c_void_p(self._stop_ptr), ctypes.c_void_p(self.stream.cuda_stream)), "mfma_load")

    def keep_alive(self):
        """queue another bounded launch behind the running one (call between passes of a long victim)"""
        self._launch()

    def __exit__(self, *exc):
        self._stop.value = 1                         # host store into coherent memory: every queued launch leaves at its first poll
        self.stream.synchronize()
        torch.cuda.current_stream().wait_stream(self.stream)
        return False


def _flat(outs):
    if torch.is_tensor(outs):
        outs = (outs,)
    return [o.detach().contiguous().clone() for o in outs if torch.is_tensor(o)]


def _same_bits(a, b):
    def raw(t):
        return t.view({4: torch.int32, 8: torch.int64, 2: torch.int16, 1: torch.uint8}[t.element_size()]) if t.dtype.is_floating_point else t
    return len(a) == len(b) and all(x.shape == y.shape and torch.equal(raw(x), raw(y)) for x, y in zip(a, b))


def _dev(a, b):
    """largest deviation of any output, relative to that output's largest magnitude.  One device->host transfer for all outputs: a victim
    with float atomics (rasteriser backward) is not bit-stable, so this runs on EVERY pass of its screen -- the per-tensor float64 copies
    and ~6 host syncs per output of the first version made such a screen take 45 s instead of 1 s."""
    rows = []
    for x, y in zip(a, b):
        if not x.dtype.is_floating_point:
            m = (x != y).any().to(torch.float32)
            rows.append(torch.stack((torch.zeros_like(m), m, torch.ones_like(m))))
            continue
        x, y = x.float(), y.float()
        fin = torch.isfinite(x) & torch.isfinite(y)
        same_nonfinite = (x == y) | (torch.isnan(x) & torch.isnan(y))
        bad = (~fin & ~same_nonfinite).any().to(torch.float32)
        zero = torch.zeros((), device=x.device)
        d = torch.where(fin, (x - y).abs(), zero).max() if x.numel() else zero
        scale = torch.where(fin, y.abs(), zero).max() if x.numel() else zero
        rows.append(torch.stack((bad, d, scale)))
    if not rows:
        return 0.0
    worst = 0.0
    for bad, d, scale in torch.stack(rows).cpu().tolist():
        if bad:
            return float("inf")
        worst = max(worst, d / scale if scale > 0 else d)
    return worst


def screen(victim, passes, kind=0, blocks=512, unloaded=3):
    """returns dict(deterministic, floor, bad, worst, passes): `bad` = passes under load that do not reproduce the unloaded result"""
    torch.cuda.synchronize()
    ref = _flat(victim())
    floor, det = 0.0, True
    for _ in range(unloaded):
        again = _flat(victim())
        if not _same_bits(again, ref):
            det = False
            floor = max(floor, _dev(again, ref))
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    load = MfmaLoad(kind, blocks)
    with load:
        for k in range(passes):
            out = _flat(victim())
            if k % 4 == 3:
                load.keep_alive()
            if det:
                ok = _same_bits(out, ref)
                d = 0.0 if ok else _dev(out, ref)
            else:
                d = _dev(out, ref)
                ok = d <= 8.0 * floor + 1e-7
            bad += int(not ok)
            worst = max(worst, d)
    torch.cuda.synchronize()
    return dict(deterministic=det, floor=floor, bad=bad, worst=worst, passes=passes)


def erratum_counters(form, kind, launches=12):
    """the self-checking packed-product kernel (csrc/selftest.hip) beside the load: (wrong low halves, wrong high halves)"""
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    ctx = MfmaLoad(kind, 512) if kind is not None else None
    if ctx is not None:
        ctx.__enter__()
    try:
        for _ in range(launches):
            capi.check(capi.lib().scp_selftest_packed_fp32(form, ctypes.c_void_p(cnt.data_ptr()), 4096, 400, capi.current_stream()), "selftest")
            torch.cuda.current_stream().synchronize()
    finally:
        if ctx is not None:
            ctx.__exit__()
    return tuple(int(x) for x in cnt.tolist())


# ---- victims --------------------------------------------------------------------------------------------------------------------------
class StepVictims:
    """every stage of one training step at the bench batch (B = 32, 256 x 256, 642 v / 1280 f) as a callable with frozen inputs"""

    def __init__(self, batch_size=8, repeat=4, device="cuda"):
        import scp_amd.dino as dino
        from scp_amd import fused_conv, synthetic
        from scp_amd.flags import Options
        from scp_amd.trainer import Trainer
        dino.ALLOW_RANDOM_INIT = True
        self.dev = device
        opts = Options("laptop_wild6d", batch_size=batch_size, repeat=repeat, train=True, ngpu=1, vis_freq=10 ** 9)
        torch.manual_seed(0)
        self.tr = tr = Trainer(opts, prior=synthetic.bottle_like(3), device=device)
        self.model = m = tr.model
        m.rotation_angle = 90.0
        n = m.mesh.symm_rots.shape[0] * opts.batch_size * opts.repeat
        g = torch.Generator().manual_seed(99)
        face_idx = torch.randint(0, m.mesh.num_faces, (n, 10000), generator=g)
        su, r2 = torch.rand(n, 10000, generator=g).sqrt(), torch.rand(n, 10000, generator=g)
        m.mesh.sample_override = (face_idx.to(device), torch.stack((1.0 - su, su * (1.0 - r2), su * r2), -1).to(device))
        self.data = synthetic.make_batch(batch_size, repeat, 256, seed=100, device=device)
        self.fused_conv = fused_conv
        # two real steps first: solver searches, lazily built planes / caches, learned unused-parameter set
        for _ in range(2):
            tr.step(self.data)
        torch.cuda.synchronize()
        self.snap_model = {k: v.detach().clone() for k, v in m.state_dict().items()}
        opt = tr.optim.optimizer
        # FlatAdamW keeps its state in flat buffers: snapshot those (one copy each) instead of a per-parameter state_dict round trip,
        # which made the `optimizer` screen spend 0.45 s per pass in restore()
        self.flat_optim = hasattr(opt, "_m") and getattr(opt, "_grads", None) is not None
        if self.flat_optim:
            self.snap_flat = (opt._m.clone(), opt._v.clone(), list(opt._steps), opt._calls, [dict((k, v) for k, v in g.items() if k != "params")
                                                                                           for g in opt.param_groups])
        self.snap_optim = None if self.flat_optim else self._clone(opt.state_dict())
        self.snap_sched = dict(tr.optim.scheduler.state_dict())
        self.iteration = tr.iteration
        self._stage_inputs()

    @staticmethod
    def _clone(obj):
        if torch.is_tensor(obj):
            return obj.detach().clone()
        if isinstance(obj, dict):
            return {k: StepVictims._clone(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(StepVictims._clone(v) for v in obj)
        return obj

    def restore(self):
        self.model.load_state_dict(self.snap_model)
        opt = self.tr.optim.optimizer
        if self.flat_optim:
            m, v, steps, calls, groups = self.snap_flat
            opt._m.copy_(m)
            opt._v.copy_(v)
            opt._steps, opt._calls = list(steps), calls
            for g, saved in zip(opt.param_groups, groups):
                g.update(saved)
        else:
            opt.load_state_dict(self._clone(self.snap_optim))
        self.tr.optim.scheduler.load_state_dict(dict(self.snap_sched))
        self.tr.iteration = self.iteration
        self.fused_conv.WEIGHT_EPOCH[0] += 1
        torch.manual_seed(7)

    def _stage_inputs(self):
        """one forward's intermediates, detached: the frozen inputs of the per-stage victims"""
        m, data = self.model, self.data
        self.restore()
        img, mask, depth, occ, center, length, foc, foc_crop, pp, pp_crop, indices, gt = data
        bsz = img.shape[0]
        self.mean_v = m.mesh.mean_v.detach()[None].expand(bsz, -1, -1)
        self.faces = m.mesh.faces[None].expand(bsz, -1, -1)
        with torch.no_grad():
            img_feat, mesh_feat, pred_v, rotation, translation, scale = m.encoder(img, self.mean_v, pp_crop, foc_crop)
        self.enc_out = tuple(t.detach().clone() for t in (img_feat, mesh_feat, pred_v, rotation, translation))
        g = torch.Generator().manual_seed(5)
        self.cot = {k: torch.randn(t.shape, generator=g).to(self.dev) for k, t in zip(("img_feat", "mesh_feat", "pred_v", "rotation", "translation"),
                                                                                  self.enc_out)}

    # -- whole step ---------------------------------------------------------------------------------------------------------------------
    def step(self):
        """Trainer.step from the same snapshot with the same seeds: 12 loss terms, clipped flat gradient, updated parameters"""
        self.restore()
        total, aux, norms = self.tr.step(self.data)
        losses = torch.stack([aux[k].detach().reshape(()) for k in sorted(aux)])
        params = torch.cat([p.detach().reshape(-1) for p in self.tr._trainable])
        return losses, self.tr.grads.flat.detach(), params, torch.stack([n.detach().reshape(()) for n in norms])

    # -- stages -------------------------------------------------------------------------------------------------------------------------
    def encoder(self):
        """image encoder forward + backward (own convolutions fwd / dgrad / wgrad, BatchNorm, stem, pooling, upsampling, jitter, heads)"""
        m, data = self.model, self.data
        self.restore()
        img, pp_crop, foc_crop = data[0], data[9], data[7]
        for p in m.encoder.parameters():
            p.grad = None
        outs = m.encoder(img, self.mean_v, pp_crop, foc_crop)[:5]
        loss = sum((o * c.reshape(o.shape)).sum() for o, c in zip(outs, self.cot.values()))
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in m.encoder.parameters() if p.grad is not None])
        return tuple(o.detach() for o in outs) + (grads,)

    def correspondence(self):
        """a7: fused feature<->vertex matching forward + backward, texture sampling"""
        m, data = self.model, self.data
        img, mask = data[0], data[1]
        img_feat, mesh_feat, pred_v = (t.clone().requires_grad_(True) for t in self.enc_out[:3])
        pointcorr, match, imatch, _ = m.corr_net.match(img_feat, mesh_feat, mask, pred_v)
        tex = m.mesh.get_texture(pred_v, self.faces, imatch, img)
        pooled = pointcorr.pooled if hasattr(pointcorr, "pooled") else pointcorr
        loss = (match * match).sum() + (imatch * imatch).sum() + (tex * tex).sum() + (pooled * pooled).sum() * 1e-3
        gi, gm = torch.autograd.grad(loss, (img_feat, mesh_feat))
        return match.detach(), imatch.detach(), tex.detach(), gi, gm

    def render_and_losses(self):
        """a2/a3/a13: the depth-group and soft-texture render passes, fused image losses, their backward down to the vertices"""
        from scp_amd import fused_losses
        m, data = self.model, self.data
        img, mask, depth, foc_crop, pp_crop = data[0], data[1], data[2], data[7], data[9]
        pred_v, rotation, translation = (t.clone().requires_grad_(True) for t in self.enc_out[2:5])
        cam = (foc_crop, pp_crop, rotation, translation)
        with torch.no_grad():
            _, match, imatch, _ = m.corr_net.match(self.enc_out[0], self.enc_out[1], mask, self.enc_out[2])
            tex = m.mesh.get_texture(self.enc_out[2], self.faces, imatch, img)
        tex = tex.clone().requires_grad_(True)
        texture = m._texture_loss(pred_v, self.faces, tex, cam, img, mask, None)
        depth_out, match_out, imatch_gt, depth_weight = m.renderer.render_depth_group(pred_v, self.faces, *cam, raw=True)
        mask_sub, depth_sub, match_sub = fused_losses.depth_group_losses(depth_out, match_out, match, depth, mask)
        loss = texture.sum() + mask_sub.sum() + depth_sub.sum() + match_sub.sum()
        grads = torch.autograd.grad(loss, (pred_v, rotation, translation, tex))
        return (texture.detach(), mask_sub.detach(), depth_sub.detach(), match_sub.detach(), depth_out.detach(), imatch_gt.detach()) + grads

    def dino_cycle(self):
        """a8/a9/a11/a12: frozen ViT keys, fused mutual nearest neighbours, vertex bridge + its backward"""
        m, data = self.model, self.data
        img, mask = data[0], data[1]
        img_feat, mesh_feat = (t.clone().requires_grad_(True) for t in self.enc_out[:2])
        pointcorr, _, _, _ = m.corr_net.match(img_feat, mesh_feat, mask, self.enc_out[2])
        dw = torch.ones(img.shape[0], m.mesh.num_verts, device=self.dev)
        out = m.pretrain_corr_net.compute_cycle_loss(img, mask, dw, pointcorr)
        gi, gm = torch.autograd.grad(out[0], (img_feat, mesh_feat))
        return out[0].detach(), out[1].detach(), out[3].detach(), gi, gm

    def rotation_cycle(self):
        """a10 + the second (half-resolution) encoder pass, forward + backward"""
        m, data = self.model, self.data
        self.restore()
        img, mask = data[0], data[1]
        for p in m.encoder.parameters():
            p.grad = None
        img_feat = self.enc_out[0].clone().requires_grad_(True)
        loss, cyc, _, _ = m.corr_net.compute_rotation_cycle_loss(img, mask, img_feat, m.encoder, angle=90.0)
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in m.encoder.parameters() if p.grad is not None])
        return loss.detach(), cyc.detach(), img_feat.grad, grads

    def regularisers(self):
        """symmetry loss (1-NN kernel), Laplacian / flatten losses, their backward"""
        m = self.model
        pred_v = self.enc_out[2].clone().requires_grad_(True)
        sym = m.mesh.compute_symmetry_loss(pred_v, self.faces)
        tri = m.triangle_loss_fn(pred_v)
        loss = sym + tri
        if self.model.opts.flatten_loss:
            loss = loss + m.flatten_loss_fn(pred_v)
        g, = torch.autograd.grad(loss, pred_v)
        return sym.detach(), tri.detach(), g

    def optimizer(self):
        """clip + NaN guard on the flat gradient buffer and the fused AdamW step (ATen multi_tensor_apply), from fixed gradients"""
        self.restore()
        tr = self.tr
        tr.grads.prepare()
        gen = torch.Generator().manual_seed(11)
        for p in tr._trainable:
            p.grad.copy_(torch.randn(p.shape, generator=gen).to(self.dev) * 1e-2)
        norms = tr.collect_grad()
        tr.optim.step(tr.iteration)
        params = torch.cat([p.detach().reshape(-1) for p in tr._trainable])
        return params, torch.stack([n.detach().reshape(()) for n in norms])


STAGES = ("encoder", "correspondence", "render_and_losses", "dino_cycle", "rotation_cycle", "regularisers", "optimizer", "step")


def raster_victims(device="cuda", batch=32, size=256):
    """function-level victims through the reference-signature boundary: every render configuration of the step, forward and backward"""
    from scp_amd import synthetic
    from scp_amd.soft_renderer import functional as srf
    v, f = synthetic.bottle_like(3)
    fv, ftex = synthetic.raster_inputs(v, f, batch, seed=0)
    fv_t, tex_t = torch.tensor(fv, device=device), torch.tensor(ftex, device=device)
    g = torch.Generator().manual_seed(1)
    cot = torch.randn(batch, 4, size, size, generator=g).to(device)
    base = dict(image_size=size, dist_func="euclidean", aggr_func_alpha="prod", aggr_func_rgb="softmax", background_color=[1, 1, 1],
                texture_type="vertex")
    out = {}
    for name, kw in (("softtex_s1e-3", dict(sigma_val=1e-3, gamma_val=1e-2)), ("depth_s1e-4", dict(sigma_val=1e-4, gamma_val=1e-4))):
        def fwd(kw=kw):
            return srf.soft_rasterize(fv_t, tex_t, **base, **kw)

        def fwd_bwd(kw=kw):
            a, b = fv_t.clone().requires_grad_(True), tex_t.clone().requires_grad_(True)
            img = srf.soft_rasterize(a, b, **base, **kw)
            ga, gb = torch.autograd.grad((img * cot).sum(), (a, b))
            return img.detach(), ga, gb
        out["raster_forward/" + name] = fwd
        out["raster_forward_backward/" + name] = fwd_bwd
    return out


if __name__ == "__main__":
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    print("co-residency screen: %s, libscp_hip = %s" % (torch.cuda.get_device_name(0), capi.LIB_PATH))
    print("erratum self-test (wrong low / high halves of 1.97e10 packed products each):")
    for form, name in ((0, "v_pk_mul_f32 op_sel:[0,1]"), (1, "v_pk_mul_f32 plain"), (2, "v_pk_mul_f32 op_sel:[1,0]")):
        print("   %-28s alone %s   beside fp32 MFMA %s   beside bf16 K=16 MFMA %s" % (
            name, erratum_counters(form, None), erratum_counters(form, 1), erratum_counters(form, 0)), flush=True)
    print("%-42s %-8s %-10s %-12s %-12s" % ("victim", "passes", "bad", "worst dev", "unloaded floor (0 = bit-identical)"))
    for name, fn in raster_victims().items():
        r = screen(fn, passes)
        print("%-42s %-8d %-10d %-12.3e %-12.3e" % (name, r["passes"], r["bad"], r["worst"], r["floor"]), flush=True)
    sv = StepVictims()
    for name in STAGES:
        r = screen(getattr(sv, name), passes if name != "step" else max(passes // 2, 10))
        print("%-42s %-8d %-10d %-12.3e %-12.3e" % ("stage/" + name, r["passes"], r["bad"], r["worst"], r["floor"]), flush=True)
