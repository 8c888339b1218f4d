"""tools/softras_ablate.py -- where does the SoftRas backward spend its time?  Runs the sigma=1e-3
texture pass backward at the bench size with profiling ablations switched on inside the kernel
(scpdbg_set_flags: 1 = no wavefront reduction / flush, 2 = no per-pair math, 3 = both = binning +
staging + bbox only)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from scp_amd import capi  # noqa: E402
from scp_amd.soft_renderer import functional as srf  # noqa: E402

lib = capi.lib()
v, f = scenes.bottle_like(3)
fv, ftex = scenes.raster_inputs(v, f, 32, seed=0)
fv_t = torch.tensor(fv, device="cuda").requires_grad_(True)
tex = torch.tensor(ftex, device="cuda").requires_grad_(True)
for name, sig, gam, rgb in (("softtex", 1e-3, 1e-2, "softmax"), ("depth", 1e-4, 1e-4, "softmax")):
    kw = dict(image_size=256, background_color=[1, 1, 1], sigma_val=sig, gamma_val=gam, dist_func="euclidean",
              aggr_func_rgb=rgb, aggr_func_alpha="prod", texture_type="vertex")
    img = srf.soft_rasterize(fv_t, tex, **kw)
    g = torch.randn_like(img)
    for flags in (0, 1, 2, 3):
        lib.scpdbg_set_flags(flags)
        for _ in range(3):
            torch.autograd.grad(img, (fv_t, tex), g, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.autograd.grad(img, (fv_t, tex), g, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        print("%-8s flags=%d backward %.3f ms" % (name, flags, e0.elapsed_time(e1) / 10))
    lib.scpdbg_set_flags(0)
