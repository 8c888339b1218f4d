"""bench.py -- train iterations/s of the self-corr-pose hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full training iteration of the reference's loop body (model/trainer.py:118-125):
MeshNet.forward (encoder, feature<->vertex correspondence, 4 SoftRas passes, DINO cycle loss,
rotation cycle loss, all losses) + backward + gradient all-reduce (N>1) + per-group clipping +
AdamW/OneCycle.  Workload = BASELINE.json's metric configuration: B = batch_size 8 x repeat 4 = 32
images of 256x256 per GPU, 642-vertex / 1280-face prior mesh ("1280" mesh), the
laptop_wild6d flag set, synthetic batch (scp_amd/synthetic.py), random-init weights (no network for the
ImageNet / DINO checkpoints).  Weak scaling: every rank processes its own 32 images; `value` counts
32-image iterations completed by all ranks per second.

A step is `Trainer.step(batch, next_data=batch)`: like `Trainer.train()`, the bench hands every step the following batch, whose
frozen-DINO pass is then enqueued on the side stream before the current step's backward (one ViT pass per step either way, DESIGN
section 5; `config.vit_lookahead` reports the unpipelined time of the same run, `--no-lookahead` times only that).

Extra JSON objects (tier contract):
  roofline      the dominant hand-written kernel family of the step = vit_gemm_kernel (csrc/vit_gemm.hip: the 37 linear
                layers of the DINO ViT with LayerNorm / GELU / residual fused).  Default main loop: fp32 products on the bf16
                matrix cores with exactly split operands (csrc/gemm_core_split.h; fp32-accurate); SCP_VIT_GEMM=fp32 selects the
                fp32 matrix cores.  bound "mfma": `achieved` = algorithmic flops of the launches (2 M N K each) / their
                duration on the kernel-duration clock; peak = 2500 / 6 = 416.7 TFLOP/s of fp32-equivalent work (six bf16 MFMA
                products per algorithmic one), or 157.3 TFLOP/s (fp32 MFMA) in fp32 mode; `vs_fp32_mfma_peak` is always there.
                `traffic` = HBM bytes per step of those launches from rocprofv3 FETCH_SIZE / WRITE_SIZE passes, read from
                profiles/r04_traffic.json when that file matches the problem size, else null.
                "others": the ViT attention kernel (mfma), the SoftRas backward of the sigma=1e-3 pass (fp32 VALU on active
                (pixel,face) pairs; HBM figure for the record) and the fused correspondence kernels, same method.
  cpu_baseline  the same step on the host CPU cores (torch CPU + the C oracle rasteriser), rank 0, N=1 only: ONE full
                B=32 training step (the bench batch itself, ~1 min), after a B=2 warm-up step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP32_VALU_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 vector peak
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix-core peak
FLOP_PER_PAIR_BWD = 220.0   # ~220 flop per active pair in the backward
FLOP_PER_PAIR_FWD = 130.0   # ... ~130 in the forward (distance, sigmoid, barycentric clip, softmax update)


class KernelTimer:
    """HIP-event timing of one native entry point on torch's current stream (the stream the C ABI launches on).  Every
    `stride`-th selected call is timed (two event records per timed launch perturb the stream they sit on: with ~45 launches
    per step wrapped, the step itself read 0.3 ms longer); `calls` counts all selected calls."""

    def __init__(self, module, name, select, stride=1, on_timed=None):
        self.module, self.name, self.select, self.stride, self.on_timed = module, name, select, stride, on_timed
        self.orig = getattr(module, name)
        self.events = []
        self.calls = 0
        self.enabled = False

    def __enter__(self):
        def wrapped(*args, **kwargs):
            if not (self.enabled and self.select(*args, **kwargs)):
                return self.orig(*args, **kwargs)
            self.calls += 1
            if self.calls % self.stride:
                return self.orig(*args, **kwargs)
            if self.on_timed:
                self.on_timed(*args, **kwargs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(*args, **kwargs)
            e1.record()
            self.events.append((e0, e1))
            return out
        setattr(self.module, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.module, self.name, self.orig)

    def mean_ms(self):
        return float(np.mean([a.elapsed_time(b) for a, b in self.events])) if self.events else None

    def total_ms(self):
        return float(np.sum([a.elapsed_time(b) for a, b in self.events])) if self.events else None


class KernelClock:
    """kernel-duration clock of the vit_linear launches (include/scp_hip.h scp_kernel_clock_*): each launch stamps its earliest
    workgroup start and latest workgroup end (100 MHz device ticks) into its own slot -- the span rocprofv3's kernel trace calls
    the launch's duration -- without events on the stream and without a sync.  `flops[i]` is the i-th launch's algorithmic work
    (None for row-selected launches, whose row count lives on the device)."""

    def __init__(self, module, name, nslots, device):
        import ctypes
        from scp_amd import capi
        self.module, self.name, self.capi, self.ctypes = module, name, capi, ctypes
        self.orig = getattr(module, name)
        self.slots = torch.empty(2 * nslots, dtype=torch.int64, device=device)
        self.slots.view(-1, 2)[:, 0] = torch.iinfo(torch.int64).max
        self.slots.view(-1, 2)[:, 1] = 0
        self.flops, self.shapes, self.enabled, self.used = [], [], False, 0

    def __enter__(self):
        def wrapped(a, w, *args, **kwargs):
            if self.enabled:
                m, k = a.shape if a is not None else (kwargs["a_planes"].rows, kwargs["a_planes"].cols)     # pre-split A operand
                self.flops.append(None if kwargs.get("rows") is not None else 2.0 * m * k * w.shape[0])
                self.shapes.append((m, w.shape[0], k))
            return self.orig(a, w, *args, **kwargs)
        setattr(self.module, self.name, wrapped)
        return self

    def start(self):
        torch.cuda.synchronize()
        self.capi.check(self.capi.lib().scp_kernel_clock_begin(self.ctypes.c_void_p(self.slots.data_ptr()), self.slots.numel() // 2), "kernel_clock_begin")
        self.enabled = True

    def stop(self):
        self.enabled = False
        self.used = self.capi.lib().scp_kernel_clock_end()

    def __exit__(self, *exc):
        setattr(self.module, self.name, self.orig)

    def result(self):
        """(sum of flops, sum of durations in ms, launches) over the full (host-counted) launches"""
        n = min(self.used, len(self.flops))
        t = self.slots.view(-1, 2)[:n].cpu().numpy()
        dur_ms = (t[:, 1] - t[:, 0]) * 1e-5                       # 100 MHz ticks -> ms
        keep = [i for i in range(n) if self.flops[i] is not None and dur_ms[i] > 0]
        if not keep:
            return None
        by_shape = {}
        for i in keep:
            e = by_shape.setdefault("M%d_N%d_K%d" % self.shapes[i], [0, 0.0, self.flops[i]])
            e[0] += 1
            e[1] += dur_ms[i]
        self.by_shape = {k: {"launches": c, "avg_launch_ms": round(t / c, 4), "TFLOPs": round(f / (t / c * 1e-3) / 1e12, 1)}
                         for k, (c, t, f) in by_shape.items()}
        return float(sum(self.flops[i] for i in keep)), float(dur_ms[keep].sum()), len(keep)


INIT_STEPS = 3
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic.json")   # written by tools/traffic_report.py from rocprofv3 PMC passes


def measured_traffic(key, size_tag):
    """HBM bytes (FETCH_SIZE with the gfx950 x2 correction + WRITE_SIZE, rocprofv3 --pmc, separate passes) recorded for
    this kernel at this problem size, or None"""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh)
        e = rec.get(key)
        return float(e["bytes"]) if e and e.get("size") == size_tag else None
    except (OSError, ValueError, KeyError):
        return None


def isolated_attention(B, n_tok, heads, hd, flops, iters=30):
    import scp_amd.dino as dino_mod
    qkv = torch.randn(B, n_tok, 3 * heads * hd, device="cuda")
    for _ in range(5):
        dino_mod.fused_attention(qkv, B, n_tok, heads, hd, hd ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        dino_mod.fused_attention(qkv, B, n_tok, heads, hd, hd ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = flops / (ms * 1e-3) / 1e12
    peak = BF16_MFMA_PEAK_TF / 6.0 if dino_mod.ATTN_MODE == "split" else FP32_VALU_PEAK_TF
    return {"avg_launch_ms": ms, "achieved": tf, "frac": tf / peak, "vs_fp32_mfma_peak": tf / FP32_VALU_PEAK_TF}


def isolated_gemms(M, C=384, iters=20):
    """the four linear layers of one ViT-S block (qkv, proj, fc1, fc2) alone on an idle device, operands as the block passes them
    (split mode with pre-split activations: qkv / fc1 / fc2 read tiled bf16 planes, proj / fc1 / fc2 write them): launch-weighted
    TFLOP/s"""
    import scp_amd.dino as dino_mod
    planes = dino_mod.GEMM_MODE == "split" and dino_mod.PRESPLIT_ACTIVATIONS and not dino_mod.MIXED_BF16
    ms_total, fl_total = 0.0, 0.0
    for name, K, N, epi in (("qkv", C, 3 * C, dino_mod.GEMM_LN), ("proj", C, C, dino_mod.GEMM_BIAS_RESIDUAL), ("fc1", C, 4 * C, dino_mod.GEMM_LN_GELU),
                            ("fc2", 4 * C, C, dino_mod.GEMM_BIAS_RESIDUAL)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        v0, v1 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        st = torch.rand(M, 2, device="cuda")
        out = torch.randn(M, N, device="cuda")
        if planes:
            attn_planes = dino_mod.QK_FROM_EPILOGUE and dino_mod.attn_mode() == "split"    # the attention then writes proj's A as planes
            a_in = None if (name == "proj" and not attn_planes) else dino_mod.split_tiled(a)
            o3 = None if name == "qkv" else dino_mod.TiledPlanes(M, N, "cuda")              # qkv feeds the attention (planes + fp32 K / V)
            w3 = dino_mod.split_weight(w) if a_in is None else dino_mod.split_tiled(w)
            qk = None
            if name == "qkv" and dino_mod.QK_FROM_EPILOGUE and dino_mod.attn_mode() == "split" and M % 1025 == 0:
                # as the block runs it: the epilogue writes the attention's Q / K planes, only the V third leaves as fp32
                qk = (dino_mod.attention_workspace(M // 1025, 1025, C // 64, "cuda"), 1025, C // 64, 0.125)
            run = lambda: dino_mod.vit_linear(a if a_in is None else None, w, v0, v1, st, None if qk else out, out=None if name == "fc1" else out,
                                              epilogue=epi, w_split=w3, a_planes=a_in, out_planes=o3, fp32_out=name != "fc1", qk_planes=qk)
        else:
            w3 = dino_mod.split_weight(w) if dino_mod.gemm_mode() == "split" else None      # bf16 / fp32 modes: vit_linear prepares W itself
            run = lambda: dino_mod.vit_linear(a, w, v0, v1, st, out, out=out, epilogue=epi, w_split=w3)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms_total += e0.elapsed_time(e1) / iters
        fl_total += 2.0 * M * N * K
    tf = fl_total / (ms_total * 1e-3) / 1e12
    qk_epi = bool(planes and dino_mod.QK_FROM_EPILOGUE and dino_mod.attn_mode() == "split" and M % 1025 == 0)
    return {"block_ms": ms_total, "achieved": tf, "frac": tf / gemm_peak_tf(), "vs_fp32_mfma_peak": tf / FP32_VALU_PEAK_TF,
            "operands": "pre-split tiled planes" if planes else "fp32 A split in registers",
            "qkv_epilogue_writes_attention_planes": qk_epi,
            "note": ("the qkv launch also does the Q / K half of the attention's re-layout pass (its epilogue splits and stores the planes): "
                     "+0.05 ms on these four launches, -0.036 ms on qkv_split_kernel per block, which is not among them (DESIGN 4.4a)") if qk_epi else ""}


def fused_conv_mode():
    from scp_amd import fused_conv
    return fused_conv.CONV_MODE


def fused_wgrad_mode():
    from scp_amd import fused_conv
    return fused_conv.WGRAD_MODE




def gemm_peak_tf():
    """peak of the ViT linear layers' main loop in algorithmic (fp32-equivalent) flops: the fp32 matrix cores' 157.3 TFLOP/s, or --
    split mode, six v_mfma_f32_32x32x16_bf16 per 16 k of an exactly split fp32 product -- a sixth of the dense bf16 peak"""
    import scp_amd.dino as dino_mod
    return BF16_MFMA_PEAK_TF / 6.0 if dino_mod.GEMM_MODE == "split" else FP32_VALU_PEAK_TF


def build_trainer(device, world, batch_size=8, repeat=4, seed=0, mixed_bf16=None, high_res=False, category=None):
    """high_res: BASELINE configs[4] geometry -- 512 x 512 images, corr_h = corr_w = 128 (the only consistent choice), ViT
    sequence 4097, icosphere-4 mesh (2562 v / 5120 f).  category: one of the five Wild6D presets (config/<cat>_wild6d/base_config.txt as
    restated in scp_amd.flags) with ITS shape prior (scp_amd.mesh.category_prior: 482 .. 995 vertices) instead of the synthetic mesh"""
    if mixed_bf16 is None:      # tools/*.py reuse this builder; SCP_MIXED_BF16=1 switches them to configs[4] precision
        mixed_bf16 = os.environ.get("SCP_MIXED_BF16", "0") == "1"
    import scp_amd.dino as dino
    from scp_amd import synthetic
    from scp_amd.flags import Options
    from scp_amd.trainer import Trainer
    dino.ALLOW_RANDOM_INIT = True
    extra = dict(img_size=512, corr_h=128, corr_w=128) if high_res else {}
    opts = Options((category or "laptop") + "_wild6d", batch_size=batch_size, repeat=repeat, train=True, ngpu=world, vis_freq=10 ** 9,
                   mixed_bf16=mixed_bf16, **extra)
    torch.manual_seed(seed)
    if category:
        from scp_amd.mesh import category_prior
        prior = category_prior(category)
    else:
        prior = synthetic.bottle_like(4 if high_res else 3)
    return Trainer(opts, prior=prior, device=device), opts


def pin_rng_consumers(model, seed=99):
    """parity runs: the step's RNG consumers get the same fixed values on both sides: colour jitter off, the
    rotation-cycle angle at 90 degrees (exact rot90 on both sides), one fixed surface sample for the symmetry loss."""
    model.encoder.random_jitter = torch.nn.Identity()
    model.rotation_angle = 90.0
    n = model.mesh.symm_rots.shape[0] * model.opts.batch_size * model.opts.repeat
    g = torch.Generator().manual_seed(seed)
    face_idx = torch.randint(0, model.mesh.num_faces, (n, 10000), generator=g)
    su, r2 = torch.rand(n, 10000, generator=g).sqrt(), torch.rand(n, 10000, generator=g)
    dev = model.mesh.mean_v.device
    model.mesh.sample_override = (face_idx.to(dev), torch.stack((1.0 - su, su * (1.0 - r2), su * r2), -1).to(dev))


def cpu_baseline(sample_bs=8, sample_repeat=4, batch_seed=100, stage_times=False):
    """the step on the host cores: torch-CPU for the stock networks, the CPU oracle (oracle/: C rasteriser,
    torch restatements of the correspondence / ViT pieces) in place of every HIP kernel
    (oracle/backend.py).  Checker code, used here only as the thing being timed for the baseline --
    the patches are undone before returning and never touch the GPU path.
    Second return value: what the parity leg (loss_delta) needs from this step -- its loss terms, predicted poses and
    discrete selections."""
    from oracle import backend as oracle_backend
    from scp_amd import synthetic as synth

    class _Patch:
        def __init__(self):
            self.saved = []

        def setattr(self, obj, name, value):
            self.saved.append((obj, name, getattr(obj, name)))
            setattr(obj, name, value)

        def undo(self):
            for obj, name, value in reversed(self.saved):
                setattr(obj, name, value)

    patch = _Patch()
    oracle_backend.install(patch)
    stages, render_calls = {}, {"fwd": [], "bwd": []}

    def timed(obj, name, key):
        """wrap obj.name so that its wall time is added to stages[key] (the CPU step is synchronous)"""
        fn = getattr(obj, name)

        def wrapped(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                stages[key] = stages.get(key, 0.0) + time.perf_counter() - t
        patch.setattr(obj, name, wrapped)

    try:
        threads = torch.get_num_threads()
        tr, _ = build_trainer("cpu", 1, 1, 2)
        tr.step(synth.make_batch(1, 2, 256, seed=0, device="cpu"))      # warm-up (allocator, oneDNN primitive caches)
        tr, _ = build_trainer("cpu", 1, sample_bs, sample_repeat)
        pin_rng_consumers(tr.model)
        data = synth.make_batch(sample_bs, sample_repeat, 256, seed=batch_seed, device="cpu")
        # per-stage seconds: forward stages by wrapping the model's own calls, the rasteriser by its entry points
        from scp_amd.soft_renderer.cuda import soft_rasterize as native
        m = tr.model
        timed(m.pretrain_corr_net.net, "key_tokens", "dino_vit_fwd")
        timed(m.encoder, "forward", "encoder_pass1_fwd")
        timed(m.corr_net, "compute_rotation_cycle_loss", "rotation_cycle_fwd (2nd encoder pass + pixel-pixel matching)")
        timed(m.corr_net, "match", "feature_vertex_match_fwd")
        timed(m.pretrain_corr_net, "compute_cycle_loss", "pretrained_cycle_fwd (incl. dino_vit_fwd)")
        for name, key in (("forward_soft_rasterize", "fwd"), ("backward_soft_rasterize", "bwd")):
            fn = getattr(native, name)

            def wrapped(*a, _fn=fn, _key=key, _sigma=9 if key == "fwd" else 12):
                t = time.perf_counter()
                out = _fn(*a)
                render_calls[_key].append((float(a[_sigma]), time.perf_counter() - t))
                return out
            patch.setattr(native, name, wrapped)
        t0 = time.perf_counter()
        tr.model.iters = tr.iteration
        tr.grads.prepare()
        total, aux = tr.model(data)
        t1 = time.perf_counter()
        total.mean().backward()
        t2 = time.perf_counter()
        tr.collect_grad()
        tr.optim.step(tr.iteration)
        t3 = time.perf_counter()
        dt = t3 - t0
        pc = tr.model.pretrain_corr_net
        ref = {"aux": {k: float(v) for k, v in aux.items()}, "total": float(total.mean()),
               "rotation": tr.model.last_pose[0].clone(), "translation": tr.model.last_pose[1].clone(),
               "geometry": tuple(t.clone() for t in tr.model.last_geometry),
               "features": tuple(t.clone() for t in tr.model.last_features),
               "nn": tuple(t.clone() for t in pc.last_nn), "topk": pc.last_topk.clone()}
        rf, rb = sum(t for _, t in render_calls["fwd"]), sum(t for _, t in render_calls["bwd"])
        stages["pretrained_cycle_fwd (excl. dino_vit_fwd)"] = stages.pop("pretrained_cycle_fwd (incl. dino_vit_fwd)") - stages.get("dino_vit_fwd", 0.0)
        stages.update({"render_fwd (%d passes)" % len(render_calls["fwd"]): rf, "render_bwd (%d passes)" % len(render_calls["bwd"]): rb,
                       "forward_total": t1 - t0, "backward_total (incl. render_bwd)": t2 - t1, "clip+adamw": t3 - t2})
        # "as written" (BASELINE.md 4.3): what the reference's own schedule would cost on these cores -- the DINO ViT over 4B images x
        # 12 blocks (timed on 8 images of the batch through the full 12-block ViT, scaled by images) instead of B x 9 1/3,
        # four render passes forward and backward instead of the deduplicated ones (a sigma = 1e-4 pass added per missing pass)
        n_img = sample_bs * sample_repeat
        with torch.no_grad():
            t = time.perf_counter()
            tr.model.pretrain_corr_net.net.model(data[0][:min(8, n_img)])
            vit12 = (time.perf_counter() - t) / min(8, n_img)
        thin_f = [t for s_, t in render_calls["fwd"] if s_ < 5e-4] or [0.0]
        thin_b = [t for s_, t in render_calls["bwd"] if s_ < 5e-4] or [0.0]
        extra = (4 * n_img * vit12 - stages.get("dino_vit_fwd", 0.0)) + max(0, 4 - len(render_calls["fwd"])) * float(np.mean(thin_f)) \
            + max(0, 4 - len(render_calls["bwd"])) * float(np.mean(thin_b))
        as_written = {"value": (n_img / 32.0) / (dt + extra), "step_s": dt + extra,
                      "dino_vit_fwd_s": 4 * n_img * vit12, "render_fwd_s (4 passes)": rf + max(0, 4 - len(render_calls["fwd"])) * float(np.mean(thin_f)),
                      "render_bwd_s (4 passes)": rb + max(0, 4 - len(render_calls["bwd"])) * float(np.mean(thin_b)),
                      "how": "deduplicated step + (4B images x 12 ViT blocks, from %d images through the full ViT) + one sigma=1e-4 pass per "
                             "render pass the build shares or skips (SURVEY F4, F5, F7, F8)" % min(8, n_img)}
    finally:
        patch.undo()
    return {"value": (n_img / 32.0) / dt, "unit": "train iters/sec (32-image iterations)", "cores": threads,
            "kind": "port", "sample": "1 full training step at B=%d (batch_size %d x repeat %d, 256x256, 642v/1280f) = the bench "
                                      "batch itself, %.1f s" % (n_img, sample_bs, sample_repeat, dt),
            "variant": "deduplicated (B unique images x 9 1/3 ViT blocks, shared render passes) -- the algorithm the GPU path runs",
            "stages_s": {k: float("%.3f" % v) for k, v in stages.items()}, "as_written": as_written}, ref


CONDITIONING_FIXTURE = os.path.join(ROOT, "tests", "golden", "step_conditioning_bottle_b8x4.npz")


def reference_band(path=CONDITIONING_FIXTURE, slack=1.5):
    """per-loss relative band RECORDED FROM THE REFERENCE at this batch size (tests/golden/make_golden.py
    step_conditioning_bottle_b32: the reference's own forward at B=32 with its encoder outputs perturbed by iid N(0, sigma^2),
    sigma in {1e-6, 3e-6, 1e-5}): band[k] = max(1e-4, slack x the largest relative deviation the reference itself shows).  Data
    file only -- nothing of tests/ is imported.  None when the fixture is absent."""
    try:
        c = np.load(path)
    except OSError:
        return None
    band = {}
    for key in c.files:
        if key.startswith("cond_"):
            k, base = key[5:], float(c["base_" + key[5:]])
            spread = float(np.abs(c[key] - base).max() / abs(base)) if base != 0 else 0.0
            band[k] = max(1e-4, slack * spread)
    return band


def pin_encoder_outputs(model, geometry, features=None):
    """Replace the VALUES of the encoder's outputs by the CPU side's, keeping the autograd path: `geometry` = (pred_v, rotation,
    translation), `features` = (img_feat, mesh_feat) or None.  With both pinned everything downstream of the encoder -- the
    feature<->vertex correspondence, the four render passes, the DINO cycle, every loss: the hot path -- sees exactly the inputs the
    reference side saw and must agree to north_star's 1e-4.  (Geometry alone is not enough to isolate it: the texture term samples
    the synthetic white-noise image at the correspondence's soft-argmax positions, which turns the encoder's 1e-6 feature rounding
    into ~1e-4 of that loss -- profiles/r04_parity_sweep.txt.)"""
    fwd = model.encoder.forward
    dev = model.mesh.mean_v.device
    pv, rot, trans = (t.to(dev) for t in geometry)
    feats = None if features is None else tuple(t.to(dev) for t in features)

    def pinned(*a, **k):
        img_feat, mesh_feat, pred_v, rotation, translation, scale = fwd(*a, **k)
        pin = lambda x, v: v.reshape(x.shape) + (x - x.detach())
        if feats is not None:
            img_feat, mesh_feat = pin(img_feat, feats[0]), pin(mesh_feat, feats[1])
        return img_feat, mesh_feat, pin(pred_v, pv), pin(rotation, rot), pin(translation, trans), scale
    model.encoder.forward = pinned


def loss_delta(ref, device, sample_bs=8, sample_repeat=4, batch_seed=100):
    """BASELINE.json's "loss delta vs ref" at the headline batch: the SAME B=32 batch, the SAME initial weights (seed 0, built on
    the host) and the same pinned RNG consumers through the first training step's forward on the GPU (HIP kernels) and on the CPU
    oracle backend (cpu_baseline's step: the oracle restatements are pinned to the reference's own recordings, tests/golden).
    The mutual-NN / top-k selections of the CPU side are injected on the GPU side.  Two legs:
      pinned        the encoder's outputs (img_feat, mesh_feat, pred_v, rotation, translation) take the CPU side's values: the
                    hot-path contract (DINO, correspondence, render, losses on identical inputs), every term must be <= 1e-4
                    relative; `pinned_geometry_only` repeats it with the features left free (the round-3 definition);
      free_running  nothing else pinned: the GPU encoder rounds pred_v / pose differently from the CPU (reported), and the
                    sigma = gamma = 1e-4 silhouette terms amplify that; each term is judged against the band the
                    REFERENCE ITSELF shows under such perturbations at this batch size (reference_band()).
    parity_ok = pinned.max_rel <= 1e-4 AND every free-running term inside its reference-recorded band AND the encoder's outputs within
    the perturbation scale that band was recorded for (1e-5); each condition is also reported on its own (parity_ok_pinned,
    parity_ok_free_running, encoder_outputs_within_band_sigma).  (The band covers the reference's response to encoder-output
    perturbations only, not the 1e-5-level per-pixel differences two legal builds of the rasteriser itself show)"""
    from scp_amd import synthetic as synth
    data = synth.make_batch(sample_bs, sample_repeat, 256, seed=batch_seed, device=device)

    def run(pinned, features=True):
        tr, _ = build_trainer(device, 1, sample_bs, sample_repeat)
        pin_rng_consumers(tr.model)
        pc = tr.model.pretrain_corr_net
        pc.nn_override = tuple(t.to(device) for t in ref["nn"])
        pc.topk_override = ref["topk"].to(device)
        if pinned:
            pin_encoder_outputs(tr.model, ref["geometry"], ref["features"] if features else None)
        tr.model.iters = 0
        with torch.no_grad():
            total, aux = tr.model(data)
        rel = {k: abs(float(v) - ref["aux"][k]) / max(abs(ref["aux"][k]), 1e-12) for k, v in aux.items()}
        out = {"rel": {k: float("%.3e" % v) for k, v in rel.items()}, "max_rel": float("%.3e" % max(rel.values())),
               "total_rel": float("%.3e" % (abs(float(total.mean()) - ref["total"]) / max(abs(ref["total"]), 1e-12)))}
        return out, tr, rel

    pinned, _, pinned_rel = run(True)
    pinned["pinned_geometry_only"] = run(True, features=False)[0]
    free, tr, free_rel = run(False)
    pv, rot, trans = (t.cpu() for t in tr.model.last_geometry)
    dev = lambda a, b: float("%.3e" % (a - b).abs().max())
    f_img, f_mesh = (t.cpu() for t in tr.model.last_features)
    free["encoder_deviation_max_abs"] = {"pred_v": dev(pv, ref["geometry"][0]), "rotation": dev(rot, ref["geometry"][1]),
                                         "translation": dev(trans, ref["geometry"][2]), "img_feat": dev(f_img, ref["features"][0]),
                                         "mesh_feat": dev(f_mesh, ref["features"][1])}
    own_bw, own_fw = tr.model.pretrain_corr_net.last_nn
    flips = float((own_bw.cpu() != ref["nn"][0]).float().mean() + (own_fw.cpu() != ref["nn"][1]).float().mean()) / 2
    band = reference_band()
    if band is not None:
        free["reference_band"] = {k: float("%.3e" % band.get(k, 1e-4)) for k in free_rel}
        free["inside_band"] = all(v <= band.get(k, 1e-4) for k, v in free_rel.items())
    else:
        free["reference_band"], free["inside_band"] = None, None
    pinned["tolerance"] = 1e-4
    pinned["ok"] = all(v <= 1e-4 for v in pinned_rel.values())
    # the conditioning fixture perturbs (pred_v, rotation, translation) with sigma up to 1e-5 (make_golden.py COND_SIGMAS): a free-running
    # encoder further than that from the CPU's is outside what the band describes
    enc_ok = all(free["encoder_deviation_max_abs"][k] <= 1e-5 for k in ("pred_v", "rotation", "translation"))
    return {"what": "first-step forward, B=%d: HIP path on the GPU vs the CPU oracle backend (identical batch, weights, pinned "
                    "jitter/angle/symmetry sample, CPU selections injected); relative per loss term" % (sample_bs * sample_repeat),
            "pinned": pinned, "free_running": free,
            # three conditions, each also reported on its own: (1) the hot path on identical inputs meets north_star's 1e-4 on every term;
            # (2) free running, every term sits inside the band the REFERENCE shows for encoder outputs perturbed at 1e-6 .. 1e-5; (3) the
            # encoder's outputs (own stem / stride-2 / 1x1 / 3x3 kernels) deviate from the CPU's by no more than that perturbation scale
            "parity_ok_pinned": bool(pinned["ok"]),
            "parity_ok_free_running": bool(free["inside_band"]) if free["inside_band"] is not None else None,
            "encoder_outputs_within_band_sigma": bool(enc_ok),
            "parity_ok": bool(pinned["ok"]) and (free["inside_band"] is not False) and bool(enc_ok),
            "max_rel": pinned["max_rel"],
            "free_running_max_rel": free["max_rel"],
            "rotation_max_abs": free["encoder_deviation_max_abs"]["rotation"],
            "translation_max_abs": free["encoder_deviation_max_abs"]["translation"],
            "mutual_nn_flip_fraction_before_injection": float("%.3e" % flips),
            "tolerance": "pinned: north_star 1e-4 relative on every term.  free_running: per term max(1e-4, 1.5 x the spread the "
                         "reference itself shows at B=32 under 1e-6..1e-5 perturbations of its encoder outputs "
                         "(tests/golden/step_conditioning_bottle_b8x4.npz)"}


def reference_kernels_same_gpu(batch=32, size=256):
    """Baseline leg, beside cpu_baseline: the REFERENCE's own SoftRas kernels (oracle/_ref: soft_rasterize_cuda_kernel.cu
    :22-671 compiled unchanged for gfx950, default FMA contraction like the authors' nvcc build, reference launch geometry)
    against libscp_hip.so on the same MI355X, the four render passes of Renderer.render_all at the bench size, forward and
    backward launch times (HIP events, 10 launches each).  None when oracle/_ref was not built (needs /root/reference)."""
    from oracle import ref_gpu
    if not ref_gpu.available():
        return None
    from scp_amd import synthetic
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    passes = {
        "mask": dict(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="hard", texture_type="surface"),
        "depth": dict(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", texture_type="vertex"),
        "softtex": dict(sigma_val=1e-3, gamma_val=1e-2, aggr_func_rgb="softmax", texture_type="vertex"),
        "hardtex": dict(sigma_val=1e-4, gamma_val=1e-3, aggr_func_rgb="hard", texture_type="vertex"),
    }
    v, f = synthetic.bottle_like(3)
    fv_np, ftex_np = synthetic.raster_inputs(v, f, batch, seed=0)
    fv = torch.tensor(fv_np, device="cuda").reshape(batch, -1, 9).contiguous()
    F_ = fv.shape[1]

    def timed(fn, iters=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    out = {}
    for name, cfg in passes.items():
        tex = (torch.ones(batch, F_, 1, 3, device="cuda") if name == "mask"
               else torch.tensor(ftex_np, device="cuda").reshape(batch, F_, 3, 3).contiguous())
        scal = ref_gpu.scalars(image_size=size, dist_func="euclidean", aggr_func_alpha="prod", **cfg)
        info = torch.zeros(batch, F_, 27, device="cuda")
        aggr = torch.zeros(batch, 2, size, size, device="cuda")
        col = torch.ones(batch, 4, size, size, device="cuda")
        gf, gt, g = torch.zeros_like(fv), torch.zeros_like(tex), torch.randn(batch, 4, size, size, device="cuda")
        row = {}
        for who, fwd, bwd in (("reference", lambda: ref_gpu.forward(fv, tex, info, aggr, col, *scal, variant="contract"),
                               lambda: ref_gpu.backward(fv, tex, col, info, aggr, gf, gt, g, *scal, variant="contract")),
                              ("own", lambda: native.forward_soft_rasterize(fv, tex, info, aggr, col, *scal),
                               lambda: native.backward_soft_rasterize(fv, tex, col, info, aggr, gf, gt, g, *scal))):
            aggr.zero_()
            row[who + "_fwd_ms"] = round(timed(fwd), 4)
            row[who + "_bwd_ms"] = round(timed(bwd), 4)
        out[name] = row
    return {"what": "reference SoftRas kernels (oracle/_ref, hipcc gfx950, reference launch geometry) vs libscp_hip.so, "
                    "B=%d %dx%d %d faces, ms per launch" % (batch, size, size, F_), **out}


def bench_posefit(args):
    """SURVEY 8f #4 beside the headline: one step = Tester.pose_fitting over a batch of 32 images (256x256, 100 RANSAC
    rounds each), inputs resident in HBM.  cpu_baseline = the numpy restatement of the reference's per-image loop
    (oracle/posefit.py) on a 4-image sample of the same batch."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    from scp_amd.synthetic import posefit_inputs
    from scp_amd import pose_fit
    B = 32
    data, _ = posefit_inputs(bsz=B, size=256, n_verts=642, seed=3)
    keys = ("depth", "mask", "match", "match_conf", "foc_crop", "pp_crop", "pred_v")
    dev_in = [data[k].cuda() for k in keys]
    fit = pose_fit.PoseFitter(img_size=256, base_rot=torch.eye(3)[None].cuda())
    for _ in range(args.warmup):
        fit.pose_fitting(*dev_in)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fit.pose_fitting(*dev_in)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pts = pose_fit.last_report["n_points"]
    out = {"metric": "pose fits/sec (256x256 images, 100 RANSAC rounds, ~%dk correspondences each)" % (sum(pts) // len(pts) // 1000),
           "value": B * args.steps / elapsed, "unit": "fits/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": "Tester.pose_fitting, B=32, 256x256"},
           "roofline": None, "cpu_baseline": None}
    if not args.no_cpu_baseline:
        from oracle import posefit as oracle_posefit      # checker code, here only as the thing being timed
        sample = 4
        host_in = [data[k][:sample].numpy() for k in keys]
        t0 = time.perf_counter()
        oracle_posefit.pose_fitting_oracle(*host_in, torch.eye(3)[None].numpy(), lambda m: torch.randint(0, m, (5,)).numpy())
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": sample / dt, "unit": "fits/sec", "cores": 1, "kind": "port",
                               "sample": "%d images of the same batch, %.2f s" % (sample, dt)}
    print(json.dumps(out))


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, one process per GPU, the way
    scripts/train.sh:5-7 / train.py:29-36 of the reference start theirs (torch.distributed.launch) -- here torch.distributed.run on
    127.0.0.1 with a free port; the ranks' stdout (rank 0 prints the one JSON line) and exit status pass through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: the only mode the host driver supports (RCCL over xGMI)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="do not hand step() the following batch: the frozen-DINO features of batch i+1 are then computed at the start "
                         "of step i+1 instead of on the side stream during step i's backward (Trainer.train() always looks ahead)")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the isolated-kernel reference figures (profiling runs: keeps extra launches out of the trace)")
    ap.add_argument("--mixed-bf16", action="store_true",
                    help="BASELINE configs[4] precision (bf16 convolutions / ViT linears, fp32 elsewhere); NOT the headline")
    ap.add_argument("--high-res", action="store_true",
                    help="BASELINE configs[4] geometry: 512x512 images, 2562-vertex / 5120-face mesh, B = --hr-batch x 4 (with --mixed-bf16: "
                         "its precision too); NOT the headline")
    ap.add_argument("--hr-batch", type=int, default=2, help="batch_size (videos) of the --high-res workload; repeat stays 4")
    ap.add_argument("--category", choices=["bottle", "bowl", "camera", "laptop", "mug"], default=None,
                    help="BASELINE configs[4] 'all 5 categories': this category's flag preset and shape prior (482-995 vertices) instead "
                         "of the laptop flags on the synthetic 642-vertex mesh; NOT the headline (tools/r06/categories.sh loops over them)")
    ap.add_argument("--workload", choices=["train", "posefit"], default="train",
                    help="train = BASELINE.json's metric (default); posefit = the test-time pose-fitting path (SURVEY 8f #4)")
    args = ap.parse_args()
    if args.workload == "posefit":
        return bench_posefit(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # SCP_DIST_BACKEND=gloo SCP_SINGLE_DEVICE=1 lets the N>1 launch path be smoke-tested on a 1-GPU box
    # (all ranks on cuda:0, gradients averaged over gloo); the real multi-GPU run uses RCCL ("nccl").
    if os.environ.get("SCP_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SCP_DIST_BACKEND", "nccl"), init_method="env://",
                                world_size=world, rank=rank)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from scp_amd import synthetic as synth
    from scp_amd.soft_renderer.cuda import soft_rasterize as native
    tr, opts = build_trainer(device, world, batch_size=args.hr_batch if args.high_res else 8, mixed_bf16=args.mixed_bf16, high_res=args.high_res,
                             category=args.category)
    if args.high_res or args.category:
        args.no_cpu_baseline = True          # the CPU leg and the parity leg are the headline workload's
    data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100 + rank, device=device)
    n_faces, n_verts = tr.model.mesh.num_faces, tr.model.mesh.num_verts

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # hand-written hot kernels timed live (HIP events on the stream each one is launched on)
    import scp_amd.dino as dino_mod
    is_softtex = lambda *a: abs(a[12] - 1e-3) < 1e-9   # sigma_val of backward_soft_rasterize(...)
    gemm_flops = []

    # row-selected launches (last block's tail on the masked tokens) are not timed: their row count lives on the device
    full_gemm = lambda a, w, *rest, **kw: kw.get("rows") is None
    def count_gemm(a, w, *rest, **kw):
        m, k = a.shape if a is not None else (kw["a_planes"].rows, kw["a_planes"].cols)
        gemm_flops.append(2.0 * m * k * w.shape[0])
    # initialisation, not measurement: the first iterations run MIOpen's solver search (cudnn.benchmark, once per
    # convolution shape and process) and fill the caching allocator -- the counterpart of a compile step.  Done
    # before the W warm-up steps so that a small --warmup still times steady-state iterations.
    # The training loop (scp_amd/trainer.py: Trainer.train, one batch of look-ahead like any input pipeline) hands step() the
    # FOLLOWING batch: its frozen-DINO features depend on nothing but the images, so that ViT pass is enqueued on the side stream
    # before this step's backward.  Per-step work is unchanged -- every step of the timed region enqueues exactly one ViT pass
    # (for the batch after it) besides its own forward / backward / optimizer -- but the pass no longer heads the critical path
    # of the step that consumes it.  The synthetic "next batch" is the same resident batch.  --no-lookahead: the unpipelined step.
    from scp_amd import streams as stream_policy
    if not stream_policy.overlap():
        args.no_lookahead = True             # SCP_STREAMS=serial: one stream, no look-ahead (scp_amd/streams.py)
    nxt = None if args.no_lookahead else data
    for _ in range(INIT_STEPS):
        tr.step(data)
    sync()
    for _ in range(args.warmup):
        tr.step(data, next_data=nxt)
    sync()
    # ---- the headline: K steps, NOTHING wrapped or instrumented (no event pairs, no kernel clock, no Python shims) ----
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step(data, next_data=nxt)
    sync()
    elapsed = time.perf_counter() - t0
    # ---- the roofline legs: the same K steps again with the hot kernels timed live (HIP events on the stream each one is launched
    # on; in-kernel duration clock for the GEMM family).  strides 5 and 3 are coprime to the 33 / 8 selected launches per step: over
    # the steps every layer shape is sampled evenly.  This loop's own wall time is reported as `instrumented_ms_per_step`.
    # the other hand-written hot kernels (VERDICT r4 missing #6): the sigma = 1e-3 forward rasteriser, the fused feature<->vertex
    # matching (forward; backward = its two kernels in one C-ABI call) and the fused mutual-NN -- timed at the C-ABI entry points
    from scp_amd import capi as capi_mod
    clib = capi_mod.lib()
    fwd_softtex = lambda *a: abs(a[9] - 1e-3) < 1e-9        # sigma_val of forward_soft_rasterize(faces, textures, faces_info, aggrs, colours, size, near, far, sigma, ...)
    anycall = lambda *a, **k: True
    with KernelTimer(native, "backward_soft_rasterize", is_softtex) as kt, \
            KernelTimer(native, "forward_soft_rasterize", fwd_softtex) as ft, \
            KernelTimer(clib, "scp_fvm_forward", anycall) as fvf, KernelTimer(clib, "scp_fvm_backward", anycall) as fvb, \
            KernelTimer(clib, "scp_mutual_nn_fused", anycall) as mnt, \
            KernelTimer(dino_mod, "fused_attention", lambda *a, **k: len(a) <= 6 and k.get("q_rows") is None, stride=3) as at, \
            KernelTimer(dino_mod, "vit_linear", full_gemm, stride=5, on_timed=count_gemm) as gt, \
            KernelClock(dino_mod, "vit_linear", 64 * args.steps + 64, device) as kc:
        kt.enabled = at.enabled = gt.enabled = ft.enabled = fvf.enabled = fvb.enabled = mnt.enabled = True
        kc.start()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(data, next_data=nxt)
        sync()
        instrumented = time.perf_counter() - t1
        kt.enabled = at.enabled = gt.enabled = ft.enabled = fvf.enabled = fvb.enabled = mnt.enabled = False
        kc.stop()
        fwd_raster_ms, fvm_fwd_ms, fvm_bwd_ms, mnn_ms = ft.mean_ms(), fvf.mean_ms(), fvb.mean_ms(), mnt.mean_ms()
        gemm_clock = kc.result()
        kernel_ms = kt.mean_ms()
        attn_ms = at.mean_ms()
        gemm_total_ms, gemm_launches, gemm_calls = gt.total_ms(), len(gt.events), gt.calls
    # the same K steps without the look-ahead (reported beside the headline, not instead of it), uninstrumented as well
    unpipelined = None
    if nxt is not None:
        tr.step(data)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            tr.step(data)
        sync()
        unpipelined = time.perf_counter() - t1

    # the GEMM family once more with the device to itself (one HIP stream for K steps): in the overlapped schedule a launch shares
    # the CUs with the encoder's and the rasteriser's kernels, so its in-schedule duration (the contract's `roofline.achieved`) says how
    # the SCHEDULE treats the kernel; this leg says what the kernel does with the whole device, on the same kernel-duration clock
    exclusive, in_schedule_ms = None, None
    if stream_policy.overlap() and world == 1 and gemm_clock:
        stream_policy.MODE = "serial"
        try:
            tr.step(data)
            sync()
            with KernelClock(dino_mod, "vit_linear", 64 * args.steps + 64, device) as kc2, \
                    KernelTimer(native, "backward_soft_rasterize", is_softtex) as kt2, \
                    KernelTimer(native, "forward_soft_rasterize", fwd_softtex) as ft2, \
                    KernelTimer(dino_mod, "fused_attention", lambda *a, **k: len(a) <= 6 and k.get("q_rows") is None, stride=3) as at2, \
                    KernelTimer(clib, "scp_fvm_forward", anycall) as fvf2, KernelTimer(clib, "scp_fvm_backward", anycall) as fvb2, \
                    KernelTimer(clib, "scp_mutual_nn_fused", anycall) as mnt2:
                for t_ in (kt2, ft2, at2, fvf2, fvb2, mnt2):
                    t_.enabled = True
                kc2.start()
                for _ in range(args.steps):
                    tr.step(data)
                sync()
                kc2.stop()
                exclusive = kc2.result()
                exclusive_by_shape = getattr(kc2, "by_shape", None)
                # the other hand-written kernels on the same footing: their in-schedule averages are kept beside these
                in_schedule_ms = {"raster_backward": kernel_ms, "raster_forward": fwd_raster_ms, "vit_attention": attn_ms,
                                  "fvm_forward": fvm_fwd_ms, "fvm_backward": fvm_bwd_ms, "mutual_nn_fused": mnn_ms}
                kernel_ms, fwd_raster_ms, attn_ms = kt2.mean_ms() or kernel_ms, ft2.mean_ms() or fwd_raster_ms, at2.mean_ms() or attn_ms
                fvm_fwd_ms, fvm_bwd_ms, mnn_ms = fvf2.mean_ms() or fvm_fwd_ms, fvb2.mean_ms() or fvm_bwd_ms, mnt2.mean_ms() or mnn_ms
        finally:
            stream_policy.MODE = "overlap"

    # a step whose gradients contain a NaN is "trained" with all-zero gradients (the reference's guard, trainer.py:140-150): the
    # timing would not notice.  The last timed step's guard flag and clip norms go into the line.
    last_clip = getattr(tr, "last_clip", None)
    grads_ok = None if last_clip is None else {"all_finite": bool(float(last_clip[6]) == 1.0),
                                                "group_norms_mean_v_shapenerf_pose": [float(x) for x in last_clip[:3]]}

    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        B, S = opts.batch_size * opts.repeat, opts.img_size
        size_tag = "B%d_S%d_V%d" % (B, S, n_verts)
        others = {}
        if kernel_ms:
            # algorithmic bytes of one softtex backward launch: faces + textures + faces_info in,
            # soft_colors + aggrs_info + grad_soft_colors in, grad_faces + grad_textures out
            alg_bytes = 4.0 * B * (n_faces * (9 + 9 + 27) + S * S * (4 + 2 + 4) + n_faces * (9 + 9))
            pairs = None
            try:  # pairs_active of this batch for the VALU-side figure
                with torch.no_grad():
                    from scp_amd.losses import project_for_render
                    m = tr.model
                    mean_v = m.mesh.mean_v[None].expand(B, -1, -1)
                    _, _, pred_v, rot, trans, _ = m.encoder(data[0], mean_v, data[9], data[7])
                    pv = project_for_render(pred_v, data[7], data[9], rot, trans)
                    pv = torch.stack((pv[..., 0], pv[..., 1], pv[..., 2] + 2.7320508), -1)
                    fv = pv[:, m.mesh.faces].reshape(B, n_faces, 9).contiguous()
                    pairs = native.count_pairs(fv, S, 1e-3, float(np.log(1. / 1e-4 - 1.)))
            except Exception:  # instrumentation only
                pairs = None
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
            raster = {"kernel": "raster_backward_kernel<softmax,vertex> (sigma=1e-3 texture pass)",
                      "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic("raster_backward", size_tag),
                      "avg_launch_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes}
            if pairs:
                tf = pairs * FLOP_PER_PAIR_BWD / (kernel_ms * 1e-3) / 1e12
                raster["valu"] = {"pairs_active": pairs, "flop_per_pair": FLOP_PER_PAIR_BWD, "achieved_TFLOPs": tf,
                                  "peak_TFLOPs": FP32_VALU_PEAK_TF, "frac": tf / FP32_VALU_PEAK_TF}
            others["raster_backward"] = raster
            if fwd_raster_ms:
                # forward of the same pass: faces + textures in, faces_info + aggrs_info + soft_colors out
                fb = 4.0 * B * (n_faces * (9 + 9 + 27) + S * S * (4 + 2))
                fr = {"kernel": ("raster_forward_pq_kernel (per-wavefront pair queue, opt-in)" if os.environ.get("SCP_RASTER_FWD") == "pq" else
                                 "raster_forward_kernel<softmax,vertex>") + " (sigma=1e-3 texture pass; + the per-face setup kernel of the same call)",
                      "bound": "hbm", "achieved": fb / (fwd_raster_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": fb / (fwd_raster_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": measured_traffic("raster_forward_softtex", size_tag),
                      "avg_launch_ms": fwd_raster_ms, "algorithmic_bytes_per_launch": fb,
                      "note": "not HBM-bound in any honest accounting: the VALU figure on active pairs is the one that grades it"}
                if pairs:
                    tf = pairs * FLOP_PER_PAIR_FWD / (fwd_raster_ms * 1e-3) / 1e12
                    fr["valu"] = {"pairs_active": pairs, "flop_per_pair": FLOP_PER_PAIR_FWD, "achieved_TFLOPs": tf,
                                  "peak_TFLOPs": FP32_VALU_PEAK_TF, "frac": tf / FP32_VALU_PEAK_TF}
                others["raster_forward"] = fr
        if fvm_fwd_ms and fvm_bwd_ms:
            P, C = opts.corr_h * opts.corr_w, opts.n_corr_feat
            prod = 2.0 * B * P * n_verts * C                      # one K = 64 score / gradient product over the [P, V] tile grid
            io_f = 4.0 * B * (C * P + n_verts * C + P + n_verts * 3 + (P // 4) * n_verts + P * 3 + 2 * n_verts + 2 * P + 2 * n_verts)
            io_b = 4.0 * B * (2 * C * P + 2 * n_verts * C + (P // 4) * n_verts + P * 3 + 2 * n_verts + 2 * P + 2 * n_verts)
            for tag, ms, nprod, io, what in (("fvm_forward", fvm_fwd_ms, 1, io_f, "fvm_forward_kernel + column merge (a7 forward: scores, both softmaxes, soft-argmaxes, 2x2 pooling)"),
                                             ("fvm_backward", fvm_bwd_ms, 3, io_b, "fvm_backward_img_kernel + fvm_backward_mesh_kernel (one C-ABI call; algorithmic = score tile once + "
                                                                                  "two gradient products, executed = the tile twice)")):
                tf = nprod * prod / (ms * 1e-3) / 1e12
                others[tag] = {"kernel": what, "bound": "mfma", "achieved": tf, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                               "frac": tf / FP32_VALU_PEAK_TF, "traffic": measured_traffic(tag, size_tag), "avg_launch_ms": ms,
                               "algorithmic_flops_per_launch": nprod * prod, "algorithmic_bytes_per_launch": io,
                               "hbm_frac_of_algorithmic_bytes": io / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "note": "fp32 MFMA (v_mfma_f32_32x32x2_f32) tiles that never leave registers; VALU-bound by the two softmaxes "
                                       "(~30 VALU instructions per score), DESIGN 4.3"}
        if mnn_ms:
            from scp_amd.losses import pair_indices
            n_pairs = int(pair_indices(tr.model.pretrain_corr_net.divide_kind, opts.batch_size, opts.repeat)[0].shape[0])
            ptok, kdim = (S // 8) ** 2, 384                       # image pairs of divide_fn (loss_utils.py:326-345), 32 x 32 DINO tokens each
            fl = 2.0 * n_pairs * ptok * ptok * kdim
            split = dino_mod.GEMM_MODE == "split"
            pk = BF16_MFMA_PEAK_TF / 6.0 if split else FP32_VALU_PEAK_TF
            tf = fl / (mnn_ms * 1e-3) / 1e12
            others["mutual_nn_fused"] = {"kernel": "mutual_nn_fused_kernel (a8: DINO-key score GEMM + both argmax reductions, no score tensor)",
                                         "bound": "mfma", "achieved": tf, "peak": pk, "unit": "TFLOP/s", "frac": tf / pk,
                                         "vs_fp32_mfma_peak": tf / FP32_VALU_PEAK_TF, "traffic": measured_traffic("mutual_nn_fused", size_tag),
                                         "avg_launch_ms": mnn_ms, "algorithmic_flops_per_launch": fl, "pairs": n_pairs}
        if attn_ms:
            n_tok, heads, hd = (S // 8) ** 2 + 1, 6, 64
            flops = 4.0 * B * heads * n_tok * n_tok * hd
            tf = flops / (attn_ms * 1e-3) / 1e12
            attn_split = dino_mod.ATTN_MODE == "split"
            attn_peak = BF16_MFMA_PEAK_TF / 6.0 if attn_split else FP32_VALU_PEAK_TF
            others["vit_attention"] = {
                "kernel": ("vit_attention_split_kernel + qkv re-layout + tail queries (flash attention on the bf16 matrix cores with exactly "
                           "split operands, N=%d, 6x64, B=%d)" if attn_split else
                           "vit_attention_kernel (fp32 MFMA flash attention, N=%d, 6x64, B=%d)") % (n_tok, B), "bound": "mfma",
                "achieved": tf, "peak": attn_peak, "unit": "TFLOP/s", "frac": tf / attn_peak, "vs_fp32_mfma_peak": tf / FP32_VALU_PEAK_TF,
                "traffic": (lambda t: None if t is None else t / 9.0)(measured_traffic("vit_attention", size_tag)),   # 8 full launches + the query-selected one of block 8
                "traffic_per_step": measured_traffic("vit_attention", size_tag), "avg_launch_ms": attn_ms,
                "algorithmic_flops_per_launch": flops, "launches_per_step": 8,      # + 1 query-selected launch (block 8), not timed
                # the live figure is taken while the encoder / render streams share the device; the same kernel alone on
                # an idle device, for reference (not the roofline claim):
                "isolated": None if args.no_isolated else isolated_attention(B, n_tok, heads, hd, flops)}
        if in_schedule_ms:
            for k_, v_ in others.items():
                v_["measured"] = "K training steps on ONE HIP stream (the kernel owns the device), HIP events around the launch"
                v_["in_schedule_avg_launch_ms"] = in_schedule_ms.get(k_)
        roofline = None
        if gemm_total_ms:
            fl = float(np.sum(gemm_flops[-gemm_launches:]))
            tf_events = fl / (gemm_total_ms * 1e-3) / 1e12
            # the contract number is taken on the KERNEL-DURATION clock (first workgroup start to last workgroup end, stamped
            # inside every launch of the timed region): it is the clock of the committed rocprofv3 kernel trace, so the figure can
            # be recomputed from profiles/.  HIP events around a launch (kept as `events`) also contain the time the launch waits
            # for CUs held by the other streams' kernels.
            if gemm_clock:
                cfl, cms, cn = gemm_clock
                tf = cfl / (cms * 1e-3) / 1e12
            else:
                cfl, cms, cn = fl, gemm_total_ms, gemm_launches
                tf = tf_events
            per_step = gemm_calls / args.steps
            split = dino_mod.GEMM_MODE == "split"
            peak = gemm_peak_tf()
            roofline = {"kernel": "vit_gemm_kernel family (%s + fused LayerNorm / GELU / bias+residual epilogues; "
                                  "%d launches per step, M = %d tokens)" % (
                                      "fp32 GEMM on the bf16 matrix cores with exactly split operands (3 bf16 terms per fp32 value, "
                                      "6 partial products, fp32 accumulation)" if split else "fp32 MFMA GEMM",
                                      per_step, B * ((S // 8) ** 2 + 1)),
                        "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                        "peak_note": ("algorithmic fp32 flops (2 M N K); peak = dense bf16 MFMA peak 2500 / 6 executed products per "
                                      "algorithmic one; executed matrix-core flops are 6x `achieved`" if split else
                                      "algorithmic fp32 flops (2 M N K) against the fp32 MFMA peak"),
                        "vs_fp32_mfma_peak": tf / FP32_VALU_PEAK_TF,
                        "arithmetic": ("fp32-accurate: operands represented exactly, dropped partial products < 2^-24 |a b|; error "
                                       "vs float64 not above the fp32 matrix cores' (tests/test_vit_gpu.py); SCP_VIT_GEMM=fp32 "
                                       "selects the fp32 cores" if split else "v_mfma_f32_32x32x2_f32"),
                        # the contract's `traffic` is per launch, like `achieved`: the family's counter bytes of one step / its launches
                        "traffic": (lambda t: None if t is None else t / per_step)(measured_traffic("vit_gemm", size_tag)),
                        "traffic_per_step": measured_traffic("vit_gemm", size_tag),
                        "traffic_source": "profiles/r06_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 "
                                          "(gfx950) + WRITE, per training step (steps counted from the trace), here / launches per step",
                        "clock": "in-kernel s_memrealtime stamps: first workgroup start to last workgroup end of every full launch "
                                 "of the timed region (= rocprofv3 kernel-trace duration; profiles/r04_kernel_stats_timed_window.csv)",
                        "avg_launch_ms": cms / cn, "algorithmic_flops_per_launch": cfl / cn,
                        "by_shape": getattr(kc, "by_shape", None),
                        "profile_note": "under rocprofv3 the step is host-bound (per-launch interception), so kernels of different streams "
                                        "hardly overlap any more and each reads much SHORTER than in the free-running step this clock "
                                        "measures (round 3: 214 vs 361 us per launch, -41 %; profiles/r04_kernel_stats_timed_window.csv "
                                        "for this round): a trace reproduces `isolated`, not the in-step figure, whose extra time is CU "
                                        "sharing with the encoder's streams, not kernel quality",
                        "launches_per_step": per_step, "timed_launches": cn,
                        "ms_per_step": cms / cn * per_step,
                        "events": {"what": "HIP events on the ViT stream around every 5th launch (includes waiting for CUs)",
                                   "achieved": tf_events, "frac": tf_events / peak,
                                   "avg_launch_ms": gemm_total_ms / gemm_launches, "timed_launches": gemm_launches},
                        "sustained_clock_note": ("dense bf16 MFMA on these operands runs the part at 1.4-1.7 GHz (power limit; "
                                                 "tools/probes/gemm_split.hip: s_memtime / s_memrealtime per kernel), i.e. 100 % "
                                                 "matrix-pipe occupancy delivers ~250-290 TFLOP/s of fp32-equivalent work; peak "
                                                 "above is the nominal 2.4 GHz figure" if split else
                                                 "fp32 MFMA on random data runs the part at ~1.95 GHz (tools/probes/gemm_v3.hip: "
                                                 "s_memtime / s_memrealtime), i.e. 128 TFLOP/s is what 100 % matrix-pipe occupancy "
                                                 "delivers; peak above is the nominal 2.4 GHz figure"),
                        # per block: qkv (r 384, w 1152), proj (r 384 + 384 residual, w 384), fc1 (r 384, w 1536), fc2 (r 1536 + 384,
                        # w 384) floats per token = 6912; + block 9's K slice (r 384, w 384); + the weights once per launch
                        "algorithmic_bytes_per_step": 4.0 * (B * ((S // 8) ** 2 + 1) * (9 * 6912 + 768) + 9 * 4608 * 384 + 384 * 384),
                        # the live figure is taken while the encoder / render streams share the device; the four layer
                        # shapes alone on an idle device, for reference (not the roofline claim):
                        "exclusive_device": None if not exclusive else {
                            "what": "the same launches of the same training step on ONE stream (SCP_STREAMS=serial for these K steps): "
                                    "nothing shares the CUs with a GEMM launch; kernel-duration clock",
                            "achieved": exclusive[0] / (exclusive[1] * 1e-3) / 1e12, "frac": exclusive[0] / (exclusive[1] * 1e-3) / 1e12 / peak,
                            "launches": exclusive[2], "avg_launch_ms": exclusive[1] / exclusive[2], "by_shape": exclusive_by_shape},
                        "isolated": None if args.no_isolated else isolated_gemms(B * ((S // 8) ** 2 + 1)),
                        "others": others}
            if exclusive:
                # Which duration grades the KERNEL?  In the overlapped schedule a GEMM launch shares the CUs with the encoder's and the
                # rasteriser's kernels: its duration there (2-2.7x longer) grades the schedule.  The contract figure -- algorithmic
                # flops / average launch duration, measured live in K training steps, reproducible from the committed rocprofv3 trace
                # (where per-launch interception makes kernels of different streams hardly overlap) -- is therefore taken from the K
                # steps run on ONE stream; the headline schedule's own durations stay beside it as `in_schedule`.
                ex_tf = exclusive[0] / (exclusive[1] * 1e-3) / 1e12
                roofline["in_schedule"] = {
                    "what": "the same launches inside the K timed-schedule steps (SCP_STREAMS=overlap: CUs shared with other streams' kernels)",
                    "achieved": roofline["achieved"], "frac": roofline["frac"], "avg_launch_ms": roofline["avg_launch_ms"],
                    "ms_per_step": roofline["ms_per_step"], "by_shape": roofline["by_shape"], "timed_launches": roofline["timed_launches"],
                    "events": roofline.pop("events")}
                ex_avg = exclusive[1] / exclusive[2]
                roofline.update(achieved=ex_tf, frac=ex_tf / peak, vs_fp32_mfma_peak=ex_tf / FP32_VALU_PEAK_TF, avg_launch_ms=ex_avg,
                                algorithmic_flops_per_launch=exclusive[0] / exclusive[2], by_shape=exclusive_by_shape,
                                timed_launches=exclusive[2], gemm_ms_per_step_one_stream_leg=ex_avg * per_step,
                                schedule_of_contract_figures="one-stream leg (K extra steps, SCP_STREAMS=serial): achieved / frac / "
                                                             "avg_launch_ms here are NOT durations inside the timed overlapped step -- "
                                                             "those are under in_schedule; the line's top-level value / ms_per_step "
                                                             "are the overlapped schedule's",
                                measured="K further training steps of the same workload on ONE HIP stream (nothing shares the CUs with "
                                         "a launch); in-kernel duration clock; agrees with the rocprofv3 kernel trace under profiles/",
                                profile_note="a rocprofv3 trace makes the step host-bound, so kernels of different streams hardly overlap "
                                             "in it: the trace reproduces THIS figure; `in_schedule` is what the free-running overlapped "
                                             "step shows for the same launches")
                del roofline["exclusive_device"]
                del roofline["ms_per_step"]      # no key of the timed step's name carrying another schedule's figure
        elif others:
            roofline = dict(next(iter(others.values())), others=others)
        out = {
            "metric": ("train iters/sec (batch=%d, %dx%d, %d-face/%d-vert mesh%s; configs[4], not the headline)" % (
                B, S, S, n_faces, n_verts, ", %s_wild6d preset + prior" % args.category if args.category else "")) if (args.high_res or args.category)
            else "train iters/sec (batch=32, 256x256, 1280-face/642-vert mesh)",
            "value": world * args.steps / elapsed, "unit": "iters/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "instrumented_ms_per_step": 1000.0 * instrumented / args.steps,
            "scaling": "weak", "vs_baseline": None, "gradients": grads_ok,
            "dtype": "bf16 convolutions + ViT linears, f32 elsewhere (configs[4] precision; not the headline)" if args.mixed_bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": ("configs[4] %s: B=%d (batch_size %d x repeat 4) %dx%d per GPU, %dv/%df mesh, %s_wild6d flags, "
                                    "full training step (fwd+bwd+clip+AdamW)" % (
                                        "geometry" if args.high_res else "category at the headline geometry", B, opts.batch_size, S, S, n_verts,
                                        n_faces, args.category or "laptop")) if (args.high_res or args.category) else
                                   "configs[2]: B=32 (batch_size 8 x repeat 4) 256x256 per GPU, 642v/1280f mesh, "
                                   "laptop_wild6d flags, full training step (fwd+bwd+clip+AdamW)",
                       "images_per_sec": world * args.steps * B / elapsed, "parallelism": "dp%d" % world,
                       "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else 0,
                       "gradient_buckets": len(tr.grads.buckets), "buckets_launched_inside_backward": tr.grads.launched_in_backward,
                       "streams": {"mode": stream_policy.MODE,
                                   "what": "overlap (the default) = frozen ViT / rotation-cycle encoder pass / texture pass on side streams + "
                                           "ViT look-ahead + gradient buckets reduced inside backward; serial = one HIP stream.  Round 4's "
                                           "co-residency hazard is a gfx950 erratum of one instruction form (packed fp32 with op_sel [0,1] "
                                           "beside K-doubled 16-bit MFMAs, profiles/r05_packed_fp32_erratum.txt): no shipped kernel and no "
                                           "ATen kernel of the step carries it (tests/test_capi_symbols.py, profiles/r05_torch_kernel_scan.txt) "
                                           "and tests/test_coresidency_gpu.py screens every stage of the step under a bf16-MFMA load"},
                       "vit_lookahead": {"enabled": not args.no_lookahead,
                                         "what": "step(data, next_data): the frozen-DINO ViT pass of the NEXT batch runs on the side stream "
                                                 "during this step's backward, as in Trainer.train(); one ViT pass per timed step either way",
                                         "unpipelined_ms_per_step": None if unpipelined is None else 1000.0 * unpipelined / args.steps,
                                         "unpipelined_iters_per_sec": None if unpipelined is None else world * args.steps / unpipelined},
                       "matrix_cores": {"vit_linear": dino_mod.GEMM_MODE, "vit_attention": dino_mod.ATTN_MODE,
                                        "encoder_conv_fwd_dgrad": fused_conv_mode(), "encoder_conv_wgrad": fused_wgrad_mode(),
                                        "note": "split = bf16 MFMA on exactly split fp32 operands, fp32 accumulate (fp32-accurate, "
                                                "DESIGN 4.4c); fp32 = v_mfma_f32_32x32x2_f32 (SCP_VIT_GEMM / SCP_VIT_ATTN / "
                                                "SCP_CONV_GEMM / SCP_CONV_WGRAD = fp32); the 7x7 stem runs on the fp32 matrix "
                                                "cores in both modes (csrc/conv_stem.hip)"}},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref_step = cpu_baseline()
            try:
                out["loss_delta"] = loss_delta(ref_step, device)
            except Exception as e:  # noqa: BLE001 -- reported, not hidden
                out["loss_delta"] = {"error": repr(e)}
            try:        # same-GPU brute-force baseline: the reference's own rasteriser kernels (baseline leg, not the product)
                out["cpu_baseline"]["same_gpu_reference_kernels"] = reference_kernels_same_gpu()
            except Exception as e:  # noqa: BLE001 -- a baseline figure must not take the bench line down
                out["cpu_baseline"]["same_gpu_reference_kernels"] = {"error": repr(e)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
