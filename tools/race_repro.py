"""tools/race_repro.py -- stand-alone reproduction of the run-to-run deviation tools/first_step_flake.py found: the rasteriser forward
on the main stream, same inputs every time, while ONE kind of ViT work runs on a side stream.  Prints, per kind, how many of the
rasteriser's outputs differ from the first (bit-exact hash).  Kinds: none, block (a whole ViT block, fused path), attn, gemm (the
four linears), stats (LayerNorm statistics), conv (the 8x8 / 8 patch embedding), cat (torch.cat + add on [32,1025,384])."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import dino  # noqa: E402
from scp_amd import synthetic as scenes  # noqa: E402
from scp_amd import corr_ops, fused_conv  # noqa: E402
from scp_amd.soft_renderer import functional as srf  # noqa: E402
from scp_amd.soft_renderer.cuda import soft_rasterize as native  # noqa: E402

dev = "cuda"
B, S = 32, 256
v, f = scenes.bottle_like(3)
fv, ftex = scenes.raster_inputs(v, f, B, seed=0)
fv_t, tex = torch.tensor(fv, device=dev), torch.tensor(ftex, device=dev)
kw = dict(image_size=S, dist_func="euclidean", aggr_func_alpha="prod", sigma_val=1e-3, gamma_val=1e-2, aggr_func_rgb="softmax",
          background_color=[1, 1, 1], texture_type="vertex")
hx = lambda t: t.contiguous().view(torch.int32).long().sum()

torch.manual_seed(0)
vit = dino.vit_small().to(dev).eval()
b, n, c = 32, 1025, 384
x0 = torch.randn(b * n, c, device=dev) * 0.1
img = torch.rand(b, 3, 256, 256, device=dev)
blk = vit.blocks[0]
qkv0 = torch.randn(b, n, 3 * c, device=dev)
wmm = torch.randn(c, 1536, device=dev)
side = torch.cuda.Stream()
ew_in = torch.randn(32, 4, 256, 256, device=dev)
_g = torch.Generator().manual_seed(3)
f_img = F.normalize(torch.randn(32, 64, 4096, generator=_g), 2, 1).to(dev)
f_mesh = F.normalize(torch.randn(32, 642, 64, generator=_g), 2, 2).to(dev)
f_verts = torch.randn(32, 642, 3, generator=_g).to(dev)
f_mask = (torch.rand(32, 4096, generator=_g) > 0.4).float().to(dev)
f_grid = (torch.rand(2, 4096, generator=_g) * 2 - 1).to(dev)
VICTIM = os.environ.get("VICTIM", "raster")
import ctypes  # noqa: E402
MFMA_KINDS = {"mfma_bf16_32": 0, "mfma_bf16_16": 1, "mfma_f16_32": 2, "mfma_f32": 3, "valu": 4, "mfma_bf16_slow": 5, "mfma_bf16_half": 6}
MFMA_ITERS = {"mfma_bf16_32": 4000, "mfma_bf16_16": 8000, "mfma_f16_32": 4000, "mfma_f32": 2000, "valu": 2000, "mfma_bf16_slow": 500, "mfma_bf16_half": 2000}      # ~0.1-0.3 ms per launch
_aggr_path = os.path.join(ROOT, "tools", "probes", "libmfma_aggr.so")
AGGR = ctypes.CDLL(_aggr_path) if os.path.exists(_aggr_path) else None
aggr_out = torch.empty(2048 * 256, device=dev)
P = dict(fv=fv_t.reshape(B, -1, 9).contiguous(), tex=tex.reshape(B, -1, 3, 3).contiguous(), fi=torch.zeros(B, fv_t.shape[1], 27, device=dev),
         ai=torch.zeros(B, 2, S, S, device=dev), sc=torch.ones(B, 4, S, S, device=dev))

_gl = torch.Generator().manual_seed(11)
LIBV = dict(cv=torch.nn.Conv2d(128, 128, 3, padding=1).to(dev),
            xin=torch.randn(32, 128, 32, 32, generator=_gl).to(dev).contiguous(memory_format=torch.channels_last),
            src=F.normalize(torch.randn(32, 64, 1024, generator=_gl), 2, 1).to(dev), tgt=F.normalize(torch.randn(32, 64, 1024, generator=_gl), 2, 1).to(dev),
            m=(torch.rand(32, 1024, generator=_gl) > 0.4).float().to(dev), g=(torch.rand(32, 2, 1024, generator=_gl) * 2 - 1).to(dev),
            sc=torch.randn(32, 1024, 642, generator=_gl).to(dev), g2=(torch.rand(2, 1024, generator=_gl) * 2 - 1).to(dev))


def victim():
    if VICTIM == "raster":
        return srf.soft_rasterize(fv_t, tex, **kw)
    if VICTIM == "elementwise":            # a torch kernel over a tensor of the rasteriser's output size
        return torch.sin(ew_in) * 2 + 1
    if VICTIM == "small":                  # the same at 1/16 of the size (launch-bound)
        return torch.sin(ew_in[:2]) * 2 + 1
    if VICTIM == "fvm":                    # the fused correspondence forward (fp32 MFMA + exp; csrc/corr_fused.hip)
        out = corr_ops.FeatureVertexMatchFused.apply(f_img, f_mesh, f_mask, f_verts, f_grid, 10., 10., 64, 64)
        return torch.cat((out[0].reshape(-1)[:1 << 20], out[1].reshape(-1), out[2].reshape(-1)))
    if VICTIM == "div":                    # a chain of torch kernels with divisions / reciprocals / exponentials
        y = ew_in
        for _ in range(6):
            y = torch.exp(-(y * y)) / (1.0 + y.abs()) + torch.rsqrt(1.0 + y * y)
        return y
    if VICTIM == "persist":                # the rasteriser through the native boundary with ONE set of buffers reused by every call
        P["sc"].fill_(1.)
        native.forward_soft_rasterize(P["fv"], P["tex"], P["fi"], P["ai"], P["sc"], S, 1., 100., 1e-3, 1e-3, 0, 9.21, 1e-2, 1, 2, 1, True)
        return P["sc"].clone()
    if VICTIM == "fresh":                  # the same with fresh buffers every call (what SoftRasterizeFunction does)
        fi = torch.zeros(B, fv_t.shape[1], 27, device=dev)
        ai = torch.zeros(B, 2, S, S, device=dev)
        sc = torch.ones(B, 4, S, S, device=dev)
        native.forward_soft_rasterize(P["fv"], P["tex"], fi, ai, sc, S, 1., 100., 1e-3, 1e-3, 0, 9.21, 1e-2, 1, 2, 1, True)
        return sc
    if VICTIM == "f64":                    # torch kernels in double precision (conversions, max, reciprocal, fma)
        y = ew_in.double()
        for _ in range(4):
            y = torch.maximum(y * 1.0000001 + 0.5, torch.full_like(y, 1e-10)).reciprocal() + y * y
        return y.float()
    if VICTIM == "setup":                  # the rasteriser's per-face pass alone (zero-size image: the tile kernel has nothing to do)
        fi = torch.zeros(B, fv_t.shape[1], 27, device=dev)
        ai = torch.zeros(B, 2, 16, 16, device=dev)
        sc = torch.ones(B, 4, 16, 16, device=dev)
        native.forward_soft_rasterize(fv_t.reshape(B, -1, 9).contiguous(), tex.reshape(B, -1, 3, 3).contiguous(), fi, ai, sc, 16, 1., 100., 1e-3, 1e-3, 0,
                                      9.21, 1e-2, 1, 2, 1, True)
        if os.environ.get("SETUP_PART") == "fi":
            return fi
        if os.environ.get("SETUP_PART") == "sc":
            return sc
        return torch.cat((fi.reshape(-1), sc.reshape(-1)))
    if VICTIM.startswith("lib:"):          # other kernels of this library as victims (deterministic forward passes)
        k = VICTIM[4:]
        if k == "conv":                    # encoder 3x3 convolution (split main loop unless SCP_CONV_GEMM=fp32), 128 ch 32 x 32
            return fused_conv.conv_bias_leaky(LIBV["xin"], LIBV["cv"])
        if k == "pp":                      # pixel <-> pixel soft-argmax forward (fp32 MFMA; csrc/corr_pp.hip)
            return corr_ops.PixelPixelSoftArgmax.apply(LIBV["src"], LIBV["tgt"], LIBV["m"], LIBV["m"], LIBV["g"], 10.)
        if k == "cols":                    # column soft-argmax over [32,1024,642] (csrc/corr.hip)
            return corr_ops.cols_forward(LIBV["sc"], None, None, LIBV["g2"], 10.)[0]
        if k == "gemm32":                  # ViT linear on the fp32 matrix cores
            return dino.vit_linear(x0, blk.mlp.fc1.weight, blk.mlp.fc1.bias, epilogue=dino.GEMM_BIAS, mode="fp32")
        if k == "gemm":                    # ViT linear on the split (bf16) main loop
            return dino.vit_linear(x0, blk.mlp.fc1.weight, blk.mlp.fc1.bias, epilogue=dino.GEMM_BIAS, mode="split")
        if k == "stats":
            return dino.row_mean_rstd(x0, 1e-6)
        raise ValueError(k)
    if VICTIM.startswith("probe:"):        # one-ingredient victims of tools/probes/mfma_aggressor.hip: div, f64, branch, trans, fma, select, pk
        kind = {"div": 0, "f64": 1, "branch": 2, "trans": 3, "fma": 4, "select": 5, "pk": 6}[VICTIM[6:]]
        o = torch.empty(4096 * 256, device=dev)
        code = AGGR.probe_victim(kind, ctypes.c_void_p(o.data_ptr()), 4096, 600, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert code == 0, code
        return o
    raise ValueError(VICTIM)


def work(kind, reps):
    with torch.no_grad():
        for _ in range(reps):
            if kind == "block":
                x2d = x0.clone()
                x3 = dino.split_tiled(x2d) if (dino.gemm_mode() == "split" and dino.PRESPLIT_ACTIVATIONS) else None
                blk.forward_fused(x2d, b, n, x3)
            elif kind == "attn":
                dino.fused_attention(qkv0, b, n, 6, 64, 0.125)
            elif kind == "gemm":
                (wq, sq, tq), (w1, s1, t1) = blk._folded()
                st = dino.row_mean_rstd(x0, 1e-6)
                dino.vit_linear(x0, wq, sq, tq, st, epilogue=dino.GEMM_LN, w_split=blk._planes["qkv"])
                h = dino.vit_linear(x0, w1, s1, t1, st, epilogue=dino.GEMM_LN_GELU, w_split=blk._planes["fc1"])
                dino.vit_linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, epilogue=dino.GEMM_BIAS, w_split=blk._planes["fc2"])
            elif kind == "stats":
                for _k in range(8):
                    dino.row_mean_rstd(x0, 1e-6)
            elif kind == "conv":
                vit.patch_embed(img)
            elif kind == "gemm32":
                (wq, sq, tq), (w1, s1, t1) = blk._folded()
                st = dino.row_mean_rstd(x0, 1e-6)
                dino.vit_linear(x0, wq, sq, tq, st, epilogue=dino.GEMM_LN, mode="fp32")
                h = dino.vit_linear(x0, w1, s1, t1, st, epilogue=dino.GEMM_LN_GELU, mode="fp32")
                dino.vit_linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, epilogue=dino.GEMM_BIAS, mode="fp32")
            elif kind in ("planes_gemm", "planes_out", "qkv_epi", "attn_presplit", "proj_planes"):
                (wq, sq, tq), (w1, s1, t1) = blk._folded()
                sp = blk._planes
                st = dino.row_mean_rstd(x0, 1e-6)
                if kind == "planes_gemm":
                    dino.vit_linear(None, w1, s1, t1, st, epilogue=dino.GEMM_LN_GELU, w_split=sp["fc1_t"], a_planes=X3)
                elif kind == "planes_out":
                    dino.vit_linear(None, w1, s1, t1, st, epilogue=dino.GEMM_LN_GELU, w_split=sp["fc1_t"], a_planes=X3,
                                    out_planes=dino.TiledPlanes(b * n, 1536, dev), fp32_out=False)
                elif kind == "qkv_epi":
                    ws = dino.attention_workspace(b, n, 6, dev)
                    dino.vit_linear(x0, wq, sq, tq, st, epilogue=dino.GEMM_LN, w_split=sp["qkv_t"], a_planes=X3, qk_planes=(ws, n, 6, 0.125))
                elif kind == "attn_presplit":
                    dino.fused_attention(QKV_PRE.view(b, n, -1), b, n, 6, 64, 0.125, presplit=WS_PRE, out_planes=dino.TiledPlanes(b * n, c, dev))
                else:
                    xx = x0.clone()
                    dino.vit_linear(None, blk.attn.proj.weight, blk.attn.proj.bias, resid=xx, out=xx, epilogue=dino.GEMM_BIAS_RESIDUAL,
                                    w_split=sp["proj_t"], a_planes=Y3, out_planes=dino.TiledPlanes(b * n, c, dev))
            elif kind in ("conv64", "conv128", "conv512"):     # encoder 3x3 convolutions on the split (bf16 MFMA) main loop, forward
                cv, xin = CONVS[kind]
                for _k in range(4):
                    fused_conv.conv_bias_leaky(xin, cv)
            elif kind in MFMA_KINDS:               # register-only loops of ONE matrix instruction (tools/probes/mfma_aggressor.hip)
                for _k in range(4):
                    code = AGGR.mfma_aggressor(MFMA_KINDS[kind], ctypes.c_void_p(aggr_out.data_ptr()), 2048, MFMA_ITERS[kind],
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                    assert code == 0, code
            elif kind == "split":                  # fp32 -> three bf16 planes (v_cvt_pk_bf16_f32), no matrix instruction
                for _k in range(4):
                    dino.split_tiled(x0)
            elif kind == "clone":
                for _k in range(4):
                    x0.clone()
            elif kind == "mm":
                torch.mm(x0, wmm)


with torch.no_grad():
    ref_out = victim()
    ref = hx(ref_out)
    work("block", 1)
    work("gemm", 1)
    CONVS = {}
    for name, (ch, hw) in {"conv64": (64, 64), "conv128": (128, 32), "conv512": (512, 8)}.items():
        cv = torch.nn.Conv2d(ch, ch, 3, padding=1).to(dev)
        xin = torch.randn(32, ch, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
        CONVS[name] = (cv, xin)
        fused_conv.conv_bias_leaky(xin, cv)
    X3 = dino.split_tiled(x0)
    (_wq, _sq, _tq), _ = blk._folded()
    WS_PRE = dino.attention_workspace(b, n, 6, dev)
    QKV_PRE = dino.vit_linear(x0, _wq, _sq, _tq, dino.row_mean_rstd(x0, 1e-6), epilogue=dino.GEMM_LN, w_split=blk._planes["qkv_t"], a_planes=X3,
                              qk_planes=(WS_PRE, n, 6, 0.125))
    Y3 = dino.fused_attention(QKV_PRE.view(b, n, -1), b, n, 6, 64, 0.125, presplit=WS_PRE, out_planes=dino.TiledPlanes(b * n, c, dev))
torch.cuda.synchronize()
kinds = sys.argv[1:] or ["none", "block", "attn", "gemm", "gemm32", "mm", "stats", "conv"]
for kind in kinds:
    reps = {"none": 0, "block": 40, "attn": 120, "gemm": 50, "stats": 400, "conv": 300, "gemm32": 40, "mm": 300, "split": 300, "clone": 300}.get(kind, 100)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work(kind, reps)
    hashes, outs = [], []
    with torch.no_grad():
        for _ in range(int(os.environ.get("ITERS", "60"))):
            o = victim()
            hashes.append(hx(o))
            if os.environ.get("DIFFS") == "1":
                outs.append(o)
    torch.cuda.synchronize()
    bad = sum(int(h != ref) for h in hashes)
    print("   pattern: " + "".join("x" if h != ref else "." for h in hashes), flush=True)
    if outs and VICTIM == "setup":
        for o, h in zip(outs, hashes):
            if h != ref:
                d = (o != ref_out).reshape(-1)
                idx = d.nonzero().reshape(-1)
                print("   %d elements differ; index mod 27: %s; faces: %s; bad %s ref %s" % (
                    int(d.sum()), sorted(set(int(i) % 27 for i in idx[:2000])), sorted(set(int(i) // 27 for i in idx[:2000]))[:12],
                    ["%.4g" % x for x in o.reshape(-1)[idx[:6]].tolist()], ["%.4g" % x for x in ref_out.reshape(-1)[idx[:6]].tolist()]), flush=True)
    if outs and VICTIM == "raster":
        shown = 0
        for o, h in zip(outs, hashes):
            if h != ref and shown < 6:
                shown += 1
                d = (o != ref_out)
                idx = d.nonzero()
                tiles = set((int(i[0]), int(i[2]) // 16, int(i[3]) // 16) for i in idx[:4096])
                per = d.reshape(d.shape[0], d.shape[1], -1).float().mean(2)          # fraction of differing pixels per (image, channel)
                rows_ = d.any(3).float().sum(2)                                     # differing rows per (image, channel)
                hot = [(int(i), int(j), "%.2f" % float(per[i, j]), int(rows_[i, j])) for i, j in (per > 0).nonzero()][:12]
                bv = o[d][:8].tolist()
                rv = ref_out[d][:8].tolist()
                print("      (image, channel, fraction, rows): %s\n      bad %s\n      ref %s" % (hot, ["%.4f" % x for x in bv], ["%.4f" % x for x in rv]), flush=True)
                ad = (o - ref_out).abs()
                print("   %d elements differ (max abs %.3e, ref there %.4f) in %d tiles; channels %s; first: %s" % (
                    int(d.sum()), float(ad.max()), float(ref_out[d][0]), len(tiles), sorted(set(int(i[1]) for i in idx[:4096])),
                    [tuple(int(x) for x in i) for i in idx[:6]]), flush=True)
    print("victim %s, side stream: %-6s -> %d of %d outputs differ from the reference pass" % (VICTIM, kind, bad, len(hashes)), flush=True)
