"""tools/first_step_flake.py -- is the first forward of a freshly built trainer reproducible?  (bench.py loss_delta's `pinned` leg is
exactly that, and showed a run-to-run deviation of the mask / match terms in round 4.)
    python tools/first_step_flake.py ref  /tmp/ref.pt          CPU reference step (bench.cpu_baseline), saved
    python tools/first_step_flake.py run  /tmp/ref.pt [n]      n fresh trainers in THIS process, the pinned forward of each: per-term
                                                               relative deviation from the reference
Environment: FLAKE_NOISE=1 runs unrelated GEMMs on another stream during the forward (contention without shared memory);
FLAKE_ONLY=dino|cycle|tex leaves only that side stream on; FLAKE_SERIAL=1 switches the side streams off (overlap_dino / rotation_cycle / texture_pass); FLAKE_WARM=1 runs a full
training step of another trainer first (as bench.py has, by the time loss_delta runs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
if mode == "ref":
    _, ref = bench.cpu_baseline()
    torch.save(ref, path)
    sys.exit(0)

from scp_amd import synthetic as synth  # noqa: E402

ref = torch.load(path)
device = torch.device("cuda:0")
data = synth.make_batch(8, 4, 256, seed=100, device=device)
if os.environ.get("FLAKE_WARM") == "1":
    tr, _ = bench.build_trainer(device, 1, 8, 4)
    tr.step(data)
    torch.cuda.synchronize()
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    tr, _ = bench.build_trainer(device, 1, 8, 4)
    bench.pin_rng_consumers(tr.model)
    pc = tr.model.pretrain_corr_net
    pc.nn_override = tuple(t.to(device) for t in ref["nn"])
    pc.topk_override = ref["topk"].to(device)
    bench.pin_encoder_outputs(tr.model, ref["geometry"], ref["features"])
    if os.environ.get("FLAKE_SERIAL") == "1":
        tr.model.overlap_dino = tr.model.overlap_rotation_cycle = tr.model.overlap_texture_pass = False
    only = os.environ.get("FLAKE_ONLY")             # dino | cycle | tex: that side stream alone
    if only:
        tr.model.overlap_dino, tr.model.overlap_rotation_cycle, tr.model.overlap_texture_pass = only == "dino", only == "cycle", only == "tex"
    stub = os.environ.get("FLAKE_STUB")             # vit: no ViT kernels (keys = zeros); nn: no mutual-NN kernel; blocks=<n>: only n ViT blocks
    if stub == "vit":
        kt = pc.net.key_tokens
        pc.net.key_tokens = lambda img, keep=None: torch.zeros(img.shape[0], 1025, 384, device=img.device)
    elif stub == "prep":                            # no patch-embedding convolution / position embedding: constant tokens
        const_tok = torch.randn(32, 1025, 384, device=device) * 0.1
        pc.net.model.prepare_tokens = lambda x: const_tok.clone()
    elif stub == "nomiopen":                        # the real prepare_tokens with the patch-embedding convolution on ATen's own kernels
        torch.backends.cudnn.enabled = False
    elif stub == "prep+conv":                       # constant tokens, but the patch-embedding convolution still runs (result dropped)
        const_tok = torch.randn(32, 1025, 384, device=device) * 0.1

        def prep(x, _m=pc.net.model):
            _m.patch_embed(x)
            return const_tok.clone()
        pc.net.model.prepare_tokens = prep
    elif stub == "layer1":                          # keys of block 1 instead of block 9: one block's worth of ViT kernels
        pc.net.feat_layer = 1
    elif stub == "nokeep":                          # all tokens through every block: no indexed tail pass
        pc._keep_tokens = lambda mask: torch.ones(mask.shape[0], 1024, dtype=torch.bool, device=mask.device)
    elif stub == "nn":
        from scp_amd import ops as _ops
        _ops.mutual_nn_pairs = lambda keys, s_, t_, m_, tok0=1: tuple(x.clone() for x in pc.nn_override)
    if "images" in os.environ.get("FLAKE_OLD", ""):       # the paired-image gathers at the end of compute_cycle_loss, as before
        ccl = pc.compute_cycle_loss
        pc.compute_cycle_loss = lambda *a, **k: ccl(*a, with_images=True, **k)
    if "nobridge" in os.environ.get("FLAKE_OLD", ""):     # a9's column soft-argmax as its own pass over pooled[]
        tr.model.corr_net.fuse_bridge = False
    tr.model.iters = 0
    sums = {}
    if os.environ.get("FLAKE_SUMS") == "1":          # checksums of the texture path's intermediates (no sync until after the step)
        gt = tr.model.mesh.get_texture

        def get_texture(pred_v, faces, imatch, img):
            out = gt(pred_v, faces, imatch, img)
            sums["imatch"], sums["img"], sums["tex"] = imatch.double().sum(), img.double().sum(), out.double().sum()
            return out
        tr.model.mesh.get_texture = get_texture
        tl = tr.model._texture_loss

        def texture_loss(pred_v, faces, tex, cam, img, mask, occ):
            out = tl(pred_v, faces, tex, cam, img, mask, occ)
            sums["mask"] = mask.double().sum()
            return out
        tr.model._texture_loss = texture_loss
        cm = tr.model.corr_net.match

        def corr_match(*a, **k):
            out = cm(*a, **k)
            sums["pooled"], sums["match"], sums["imatch_o"] = (out[0].pooled.double().clamp(min=-10).sum(), out[1].double().sum(),
                                                               out[2].double().sum())
            return out
        tr.model.corr_net.match = corr_match
        rdg = tr.model.renderer.render_depth_group

        def render_depth_group(*a, **k):
            out = rdg(*a, **k)
            for i, t in enumerate(out):
                if torch.is_tensor(t):
                    sums["rdg%d" % i] = t.double().abs().sum()
            return out
        tr.model.renderer.render_depth_group = render_depth_group
        from scp_amd import fused_losses as _fl
        if not hasattr(_fl, "_orig_tl"):
            _fl._orig_tl = _fl.texture_loss

        def tl_sum(tex_out, img, mask):
            sums["tex_out"] = tex_out.double().abs().sum()
            return _fl._orig_tl(tex_out, img, mask)
        _fl.texture_loss = tl_sum
        from scp_amd.soft_renderer.cuda import soft_rasterize as _nat
        from scp_amd.soft_renderer import functional as _srf
        if not hasattr(_nat, "_orig_fwd"):
            _nat._orig_fwd, _nat._orig_dual = _nat.forward_soft_rasterize, _nat.forward_soft_rasterize_dual
        hx = lambda t: t.contiguous().view(torch.int32).long().sum()
        ras = []

        def fwd(faces, textures, faces_info, aggrs_info, soft_colors, *rest):
            pre = (hx(faces), hx(textures), hx(faces_info), hx(aggrs_info), hx(soft_colors))
            out = _nat._orig_fwd(faces, textures, faces_info, aggrs_info, soft_colors, *rest)
            ras.append(("s",) + pre + (hx(faces_info), hx(aggrs_info), hx(soft_colors)))
            return out

        def dual(faces, textures, faces_info, aggrs_info, soft_colors, textures_hard, aggrs_hard, hard, *rest):
            pre = (hx(faces), hx(textures), hx(textures_hard), hx(faces_info), hx(soft_colors), hx(hard))
            out = _nat._orig_dual(faces, textures, faces_info, aggrs_info, soft_colors, textures_hard, aggrs_hard, hard, *rest)
            ras.append(("d",) + pre + (hx(faces_info), hx(aggrs_info), hx(soft_colors), hx(aggrs_hard), hx(hard)))
            return out
        _nat.forward_soft_rasterize = _srf._native.forward_soft_rasterize = fwd
        _nat.forward_soft_rasterize_dual = _srf._native.forward_soft_rasterize_dual = dual
        sums["ras"] = ras
        from scp_amd import losses as _losses
        if not hasattr(_losses, "_orig_project"):
            _losses._orig_project = _losses.project_for_render
        proj_sums = []

        def project(*a, **k):
            out = _losses._orig_project(*a, **k)
            proj_sums.append(out.contiguous().view(torch.int32).long().sum())        # exact: any changed bit shows
            proj_sums.append((a[0].bmm(a[3])).contiguous().view(torch.int32).long().sum())   # verts.bmm(rotation) alone
            return out
        _losses.project_for_render = project
        from scp_amd import renderer as _renderer
        _renderer.project_for_render = project
        sums["proj"] = proj_sums
    if os.environ.get("FLAKE_NOISE") == "1":         # unrelated work on another stream (shares no memory with the step): contention only
        ns = torch.cuda.Stream(device=device)
        ns.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(ns):
            na = torch.randn(8192, 8192, device=device)
            for _ in range(40):
                nb = na @ na
                na = nb * 1e-4
    if os.environ.get("FLAKE_CANARY") == "1":
        # every torch.empty / empty_like / zeros made while the ViT pass (key_tokens) is being enqueued gets 4 KiB of canary in front and
        # behind; a kernel of the C ABI writing outside its output shows up as a damaged canary
        import sys as _sys
        PAD = 4096
        guards = []
        real_empty, real_empty_like, real_zeros = torch.empty, torch.empty_like, torch.zeros
        active = {"on": False}

        def guarded(shape, dtype, dev, zero):
            n = 1
            for d in shape:
                n *= int(d)
            nbytes = n * torch.empty((), dtype=dtype).element_size()
            raw = real_empty(nbytes + 2 * PAD, dtype=torch.uint8, device=dev)
            raw[:PAD] = 0xA5
            raw[PAD + nbytes:] = 0xA5
            body = raw[PAD:PAD + nbytes]
            if zero:
                body.zero_()
            f = _sys._getframe(2)
            guards.append((raw, nbytes, tuple(shape), str(dtype), "%s:%d < %s:%d" % (f.f_code.co_name, f.f_lineno, f.f_back.f_code.co_name,
                                                                                    f.f_back.f_lineno)))
            return body.view(dtype).reshape(shape)

        def norm_shape(a):
            return tuple(a[0]) if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else tuple(a)

        def empty(*a, dtype=None, device=None, **k):
            if not active["on"] or device is None or torch.device(device).type != "cuda" or k:
                return real_empty(*a, dtype=dtype, device=device, **k)
            return guarded(norm_shape(a), dtype or torch.float32, device, False)

        def zeros(*a, dtype=None, device=None, **k):
            if not active["on"] or device is None or torch.device(device).type != "cuda" or k:
                return real_zeros(*a, dtype=dtype, device=device, **k)
            return guarded(norm_shape(a), dtype or torch.float32, device, True)

        def empty_like(t, **k):
            if not active["on"] or not t.is_cuda or k or not t.is_contiguous():
                return real_empty_like(t, **k)
            return guarded(tuple(t.shape), t.dtype, t.device, False)
        kt0 = pc.net.key_tokens

        def key_tokens(img, keep=None):
            active["on"] = True
            try:
                return kt0(img, keep)
            finally:
                active["on"] = False
        pc.net.key_tokens = key_tokens
        torch.empty, torch.empty_like, torch.zeros = empty, empty_like, zeros
        with torch.no_grad():
            tr.model(data)
        torch.empty, torch.empty_like, torch.zeros = real_empty, real_empty_like, real_zeros
        torch.cuda.synchronize()
        bad = 0
        for raw, nbytes, shape, dt, where in guards:
            lo, hi = raw[:PAD], raw[PAD + nbytes:]
            nlo, nhi = int((lo != 0xA5).sum()), int((hi != 0xA5).sum())
            if nlo or nhi:
                bad += 1
                first_hi = int((hi != 0xA5).nonzero()[0]) if nhi else -1
                print("CANARY DAMAGED: %s %s at %s: %d bytes before, %d bytes after (first at +%d)" % (shape, dt, where, nlo, nhi, first_hi), flush=True)
        print("trainer %d: %d guarded allocations, %d damaged" % (i, len(guards), bad), flush=True)
        continue
    if os.environ.get("FLAKE_STREAMCHECK") == "1":
        # which raw pointers handed to the C ABI live in memory the caching allocator gave out on ANOTHER stream than the one the
        # launch goes to?  (shared read-only inputs are expected; an output or a workspace in that list is a hazard)
        import bisect
        import collections
        import sys as _sys
        orig_ptr = torch.Tensor.data_ptr
        seen = collections.Counter()
        index = {"addr": [], "seg": []}

        def refresh():
            segs = sorted(torch.cuda.memory_snapshot(), key=lambda g: g["address"])
            index["addr"] = [g["address"] for g in segs]
            index["seg"] = segs

        def lookup(ptr):
            for attempt in range(2):
                i = bisect.bisect_right(index["addr"], ptr) - 1
                if i >= 0 and ptr < index["seg"][i]["address"] + index["seg"][i]["total_size"]:
                    return index["seg"][i]["stream"]
                refresh()
            return None

        def data_ptr(t):
            ptr = orig_ptr(t)
            if t.is_cuda and ptr:
                cur = torch.cuda.current_stream().cuda_stream
                st = lookup(ptr)
                if st is not None and st != cur:
                    f = _sys._getframe(1)
                    names = []
                    while f is not None and len(names) < 5:
                        names.append("%s:%d" % (f.f_code.co_name, f.f_lineno))
                        f = f.f_back
                    seen[("alloc on %x, launch on %x" % (st, cur), tuple(t.shape), " < ".join(names))] += 1
            return ptr
        torch.Tensor.data_ptr = data_ptr
        with torch.no_grad():
            tr.model(data)
        torch.Tensor.data_ptr = orig_ptr
        for k, v in sorted(seen.items(), key=lambda kv: kv[0][2]):
            print("%4d x %s %s  %s" % (v, k[0], k[1], k[2]), flush=True)
        sys.exit(0)
    with torch.set_grad_enabled(os.environ.get("FLAKE_GRAD") == "1"):      # FLAKE_GRAD=1: keep the autograd graph (nothing is freed early)
        total, aux = tr.model(data)
    if sums:
        for rec in sums.pop("ras", []):
            print("   ras " + rec[0] + " " + " ".join("%d" % int(v) for v in rec[1:]), flush=True)
        proj = sums.pop("proj", [])
        print("   sums " + "  ".join("%s %.10e" % (k, float(v)) for k, v in sums.items()) + "  proj " + " ".join("%d" % int(v) for v in proj), flush=True)
    rel = {k: abs(float(v) - ref["aux"][k]) / max(abs(ref["aux"][k]), 1e-12) for k, v in aux.items()}
    print("trainer %d: " % i + "  ".join("%s %.3e" % (k.replace("_loss", ""), v) for k, v in rel.items() if v > 0), flush=True)
