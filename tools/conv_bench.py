"""tools/conv_bench.py -- the encoder's convolution layers at B=32, 256x256 (model/module/network/image_encoder.py:119-193): own
NHWC implicit-GEMM kernels (csrc/conv_igemm.hip forward / input gradient, csrc/conv_wgrad.hip weight gradient) against MIOpen
through F.conv2d / autograd (cudnn.benchmark, channels_last), device time per launch and TFLOP/s, per layer and summed per
encoder pass.  Layers the own kernels do not cover are listed with MIOpen only."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import capi, fused_conv  # noqa: E402

torch.backends.cudnn.benchmark = True
L = capi.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SPLIT = os.environ.get("SCP_CONV_GEMM", "split") == "split"
WSPLIT = os.environ.get("SCP_CONV_WGRAD", "split" if SPLIT else "fp32") == "split"     # own kernels: bf16 cores on split operands, or fp32 cores
# (name, Cin, Cout, k, stride, H_in, count per encoder pass)
LAYERS = [("stem 7x7/2", 3, 64, 7, 2, 256, 1), ("layer1 3x3", 64, 64, 3, 1, 64, 4), ("layer2.0 3x3/2", 64, 128, 3, 2, 64, 1),
          ("layer2 3x3", 128, 128, 3, 1, 32, 3), ("layer2 down 1x1/2", 64, 128, 1, 2, 64, 1), ("layer3.0 3x3/2", 128, 256, 3, 2, 32, 1),
          ("layer3 3x3", 256, 256, 3, 1, 16, 3), ("layer3 down 1x1/2", 128, 256, 1, 2, 32, 1), ("layer4.0 3x3/2", 256, 512, 3, 2, 16, 1),
          ("layer4 3x3", 512, 512, 3, 1, 8, 3), ("layer4 down 1x1/2", 256, 512, 1, 2, 16, 1),
          ("upconv5 3x3", 512, 256, 3, 1, 16, 1), ("iconv4 3x3", 512, 256, 3, 1, 16, 1), ("upconv4 3x3", 256, 128, 3, 1, 32, 1),
          ("iconv3 3x3", 256, 128, 3, 1, 32, 1), ("upconv3 3x3", 128, 64, 3, 1, 64, 1), ("iconv2 3x3", 128, 64, 3, 1, 64, 1),
          ("proj 1x1", 64, 64, 1, 1, 64, 1)]


def t(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {k: 0.0 for k in ("mf", "md", "mw", "of", "od", "ow")}
print("%-20s %3s %7s | MIOpen us fwd dgrad wgrad | own us fwd dgrad wgrad | own TFLOP/s fwd dgrad wgrad | MIOpen TFLOP/s" % ("layer", "cnt", "GFLOP"))
ONLY = os.environ.get("SCP_BENCH_ONLY")          # substring filter on the layer name (sweeps)
for name, cin, cout, k, s, h, cnt in LAYERS:
    if ONLY and ONLY not in name:
        continue
    x = torch.randn(B, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(x, w, None, s, k // 2)
    g = torch.randn_like(y)
    ho = y.shape[2]
    gf = 2.0 * B * ho * ho * cout * cin * k * k / 1e9
    mf = t(lambda: F.conv2d(x, w, None, s, k // 2))
    md = t(lambda: torch.autograd.grad(y, x, g, retain_graph=True)) if cin > 3 else float("nan")
    mw = t(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
    of = od = ow = float("nan")
    xn, wn, gn = x.detach().permute(0, 2, 3, 1), w.detach().permute(0, 2, 3, 1), g.permute(0, 2, 3, 1)      # NHWC views of the same storage
    assert xn.is_contiguous() and wn.is_contiguous() and gn.is_contiguous()
    if k == 7:
        # the stem: csrc/conv_stem.hip on the NCHW image (convolution + BatchNorm statistics in the forward launch)
        xs, wd = x.detach().contiguous(), w.detach()
        yo = torch.empty(B, ho, ho, cout, device="cuda")
        st = torch.empty(4, cout, device="cuda")
        rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        wsb = torch.empty(2 * L.scp_stem_conv_tiles(B, h, h) * cout, device="cuda")
        of = t(lambda: capi.check(L.scp_stem_conv_forward_bn(P(xs), P(wd), wd.stride(0), wd.stride(1), wd.stride(2), wd.stride(3), P(yo), B, h, h,
                                                             P(None), P(None), P(rm), P(rv), P(None), 0.1, 1e-5, P(st[0]), P(st[1]), P(st[2]), P(st[3]),
                                                             P(wsb), wsb.numel() * 4, capi.ticket(xs.device), capi.current_stream()), "stem fwd"))
        assert os.environ.get("SCP_BENCH_NOCHECK") or (yo - y.detach().permute(0, 2, 3, 1)).abs().max() <= 2e-4 * y.abs().max()
        wgb = L.scp_stem_conv_weight_grad_workspace(B, h, h)
        wg = torch.empty(wgb // 4, device="cuda")
        dw = torch.empty_like(wd)
        ow = t(lambda: capi.check(L.scp_stem_conv_weight_grad(P(xs), P(gn), P(dw), dw.stride(0), dw.stride(1), dw.stride(2), dw.stride(3), P(wg), wgb,
                                                              B, h, h, capi.current_stream()), "stem wgrad"))
        (dw_ref,) = torch.autograd.grad(y, w, g, retain_graph=True)
        assert os.environ.get("SCP_BENCH_NOCHECK") or (dw - dw_ref).abs().max() <= 2e-4 * dw_ref.abs().max()
    if k in (1, 3) and cin >= 32:
        wn3 = fused_conv.split_planes_tiled(wn) if SPLIT else None
        skf, skf_b = fused_conv._splitk(B, h, h, cin, cout, k, s, SPLIT, "cuda")
        yo = torch.empty(B, ho, ho, cout, device="cuda")
        of = t(lambda: capi.check(L.scp_conv_nhwc_forward(P(xn), P(None if SPLIT else wn), P(wn3), P(None), P(yo), P(None), B, h, h, cin, cout, k, s, 0, 0.0, P(skf), skf_b,
                                                          capi.current_stream()), "fwd"))
        assert os.environ.get("SCP_BENCH_NOCHECK") or (yo - y.detach().permute(0, 2, 3, 1)).abs().max() <= 2e-4 * y.abs().max()
        if s == 1:
            wt = w.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()
            wt3 = fused_conv.split_planes_tiled(wt) if SPLIT else None
            skd, skd_b = fused_conv._splitk(B, h, h, cout, cin, k, 1, SPLIT, "cuda")
            dxo = torch.empty(B, h, h, cin, device="cuda")
            od = t(lambda: capi.check(L.scp_conv_nhwc_forward(P(gn), P(None if SPLIT else wt), P(wt3), P(None), P(dxo), P(None), B, h, h, cout, cin, k, 1, 0, 0.0, P(skd), skd_b,
                                                              capi.current_stream()), "dgrad"))
        elif SPLIT and k == 3:
            # stride 2: the parity-class input gradient on the `dgrad` planes of scp_conv_weight_planes
            wd = w.detach()
            pf = torch.empty(fused_conv.tiled_planes_numel(cout, 9 * cin), dtype=torch.bfloat16, device="cuda")
            pd = torch.empty(fused_conv.tiled_planes_numel(cin, 9 * cout), dtype=torch.bfloat16, device="cuda")
            capi.check(L.scp_conv_weight_planes(P(wd), wd.stride(0), wd.stride(1), wd.stride(2), wd.stride(3), cout, cin, 3, P(pf), P(pd),
                                                capi.current_stream()), "planes")
            dxo = torch.empty(B, h, h, cin, device="cuda")
            od = t(lambda: capi.check(L.scp_conv_nhwc_dgrad_stride2(P(gn), P(pd), P(dxo), B, ho, ho, cout, cin, capi.current_stream()), "dgrad s2"))
            (dx_ref,) = torch.autograd.grad(y, x, g, retain_graph=True)
            assert os.environ.get("SCP_BENCH_NOCHECK") or (dxo - dx_ref.permute(0, 2, 3, 1)).abs().max() <= 2e-4 * dx_ref.abs().max()
        if (k == 3 and s == 1) or ((s == 2 or k == 1) and WSPLIT):
            ws_bytes = L.scp_conv_nhwc_weight_grad_workspace(B, h, h, cin, cout, k, s)
            if ws_bytes:
                ws = torch.empty(ws_bytes // 4, device="cuda")
                dw = torch.empty(cout, k, k, cin, device="cuda")
                ow = t(lambda: capi.check(L.scp_conv_nhwc_weight_grad(P(xn), P(gn), P(dw), P(None), P(ws), ws_bytes, B, h, h, cin, cout,
                                                                      k, s, int(WSPLIT), capi.current_stream()), "wgrad"))
    for key, v in (("mf", mf), ("md", md), ("mw", mw), ("of", of), ("od", od), ("ow", ow)):
        # a layer the own kernels do not cover counts with MIOpen's time on both sides (that is what the encoder runs)
        alt = {"of": mf, "od": md, "ow": mw}.get(key)
        v = alt if (v != v and alt is not None) else v
        if v == v:
            tot[key] += cnt * v
    tf = lambda us: gf / us * 1e3 if us == us else float("nan")
    print("%-20s %3d %7.2f | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f | %5.0f %5.0f %5.0f | %5.0f %5.0f %5.0f" % (
        name, cnt, gf, mf, md, mw, of, od, ow, tf(of), tf(od), tf(ow), tf(mf), tf(md), tf(mw)))
print("per encoder pass (ms): MIOpen fwd %.2f dgrad %.2f wgrad %.2f = %.2f | own (+MIOpen where not covered) fwd %.2f dgrad %.2f wgrad %.2f = %.2f" % (
    tot["mf"] / 1e3, tot["md"] / 1e3, tot["mw"] / 1e3, (tot["mf"] + tot["md"] + tot["mw"]) / 1e3,
    tot["of"] / 1e3, tot["od"] / 1e3, tot["ow"] / 1e3, (tot["of"] + tot["od"] + tot["ow"]) / 1e3))
