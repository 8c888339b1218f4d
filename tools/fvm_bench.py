"""tools/fvm_bench.py -- the three kernels of the fused feature <-> vertex correspondence (csrc/corr_fused.hip) one by one at the
bench size (B = 32, 64 x 64 feature map, 64 channels, V = 642): forward (+ column merge), backward w.r.t. the image features,
backward w.r.t. the mesh features.  HIP events around 20 launches each.  SCP_HIP_LIB=<variant .so> times a probe build."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import corr_ops  # noqa: E402

B, hf, wf, C = 32, 64, 64, 64
V = int(sys.argv[1]) if len(sys.argv) > 1 else 642
P = hf * wf
g = torch.Generator().manual_seed(V)
img = torch.nn.functional.normalize(torch.randn(B, C, P, generator=g), 2, 1).cuda()
mesh = torch.nn.functional.normalize(torch.randn(B, V, C, generator=g), 2, 2).cuda()
verts = torch.randn(B, V, 3, generator=g).cuda()
xs, ys = (torch.arange(float(wf)) + 0.5) / (wf / 2) - 1, (torch.arange(float(hf)) + 0.5) / (wf / 2) - 1
grid = torch.stack((xs.repeat(hf), ys.repeat_interleave(wf))).cuda()
yy, xx = torch.meshgrid(torch.arange(hf), torch.arange(wf), indexing="ij")
mask = (((yy - 32) ** 2 + (xx - 32) ** 2) < 24 ** 2).float().reshape(1, P).repeat(B, 1).cuda()       # centred object, ~44 % of the map
w_p, w_m, w_i = (torch.randn(B, P // 4, V, generator=g).cuda() * 1e-3, torch.randn(B, P, 3, generator=g).cuda(),
                 torch.randn(B, 2, V, generator=g).cuda())


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


import ctypes  # noqa: E402

from scp_amd import capi  # noqa: E402

L = capi.lib()
pooled = torch.empty(B, P // 4, V, device="cuda")
match, imatch = torch.empty(B, P, 3, device="cuda"), torch.empty(B, 2, V, device="cuda")
rowstat, colstat = torch.empty(B, P, 2, device="cuda"), torch.empty(B, V, 2, device="cuda")
nbytes = L.scp_fvm_workspace(B, hf, V)
ws = torch.empty(nbytes // 4, device="cuda")
g_img, g_mesh = torch.empty_like(img), torch.empty_like(mesh)
ptr = capi.dev_ptr


def forward():
    capi.check(L.scp_fvm_forward(ptr(img, "i"), ptr(mesh, "m"), ptr(mask, "k"), ptr(verts, "v"), ptr(grid, "g"), 10., 10., B, C, hf, wf, V,
                                 ptr(pooled, "p"), ptr(match, "m"), ptr(imatch, "i"), ptr(rowstat, "r"), ptr(colstat, "c"), None, None, None, ptr(ws, "w"),
                                 ctypes.c_size_t(nbytes), capi.current_stream()), "fwd")


gh = torch.nn.functional.interpolate(grid.reshape(2, hf, wf)[None], (hf // 2, wf // 2), mode="bilinear").reshape(2, -1).contiguous()
bxy, bstat = torch.empty(B, 2, V, device="cuda"), torch.empty(B, V, 2, device="cuda")


def forward_bridge():
    capi.check(L.scp_fvm_forward(ptr(img, "i"), ptr(mesh, "m"), ptr(mask, "k"), ptr(verts, "v"), ptr(grid, "g"), 10., 10., B, C, hf, wf, V,
                                 ptr(pooled, "p"), ptr(match, "m"), ptr(imatch, "i"), ptr(rowstat, "r"), ptr(colstat, "c"), ptr(gh, "gh"),
                                 ptr(bxy, "x"), ptr(bstat, "s"), ptr(ws, "w"), ctypes.c_size_t(nbytes), capi.current_stream()), "fwd")


def backward(gi, gm):
    capi.check(L.scp_fvm_backward(ptr(img, "i"), ptr(mesh, "m"), ptr(mask, "k"), ptr(verts, "v"), ptr(grid, "g"), 10., 10., B, C, hf, wf, V,
                                  ptr(match, "m"), ptr(imatch, "i"), ptr(rowstat, "r"), ptr(colstat, "c"), ptr(w_m, "gm"), ptr(w_i, "gi"),
                                  ptr(w_p, "gp"), capi.opt_ptr(gi, "gi"), capi.opt_ptr(gm, "gm"), capi.current_stream()), "bwd")


forward()
t_f = timed(forward)
t_fb = timed(forward_bridge)
t_i = timed(lambda: backward(g_img, None))
t_m = timed(lambda: backward(None, g_mesh))
print("fvm V=%d lib=%s: forward (+ column merge) %.3f ms (%.3f ms with the pooled scores' column soft-argmax for a9), backward_img %.3f ms, "
      "backward_mesh %.3f ms, sum %.3f ms" % (V, os.path.basename(os.environ.get("SCP_HIP_LIB", "libscp_hip.so")), t_f, t_fb, t_i, t_m,
                                              t_f + t_i + t_m))
