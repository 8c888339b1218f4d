"""tools/softras_microbench.py -- times the four render passes (forward + backward) of the HIP
rasteriser at the BASELINE size (B=32, 256^2, 642 v / 1280 f) and reports bbox-surviving pairs.
Run on the GPU box:  python tools/softras_microbench.py [--batch 32] [--size 256] [--subdiv 3]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import synthetic as scenes  # noqa: E402
from scp_amd.soft_renderer import functional as srf  # noqa: E402
from scp_amd.soft_renderer.cuda import soft_rasterize as native  # noqa: E402

PASSES = {
    "mask": dict(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="hard", background_color=[0, 0, 0], texture_type="surface"),
    "depth": dict(sigma_val=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax", background_color=[1, 1, 1], texture_type="vertex"),
    "softtex": dict(sigma_val=1e-3, gamma_val=1e-2, aggr_func_rgb="softmax", background_color=[1, 1, 1], texture_type="vertex"),
    "hardtex": dict(sigma_val=1e-4, gamma_val=1e-3, aggr_func_rgb="hard", background_color=[0, 0, 0], texture_type="vertex"),
}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--subdiv", type=int, default=3)
    args = ap.parse_args()
    v, f = scenes.bottle_like(args.subdiv)
    fv, ftex = scenes.raster_inputs(v, f, args.batch, seed=0)
    dev = "cuda"
    fv_t = torch.tensor(fv, device=dev).requires_grad_(True)
    res = {"B": args.batch, "S": args.size, "V": int(v.shape[0]), "F": int(f.shape[0])}
    dist_eps = float(np.log(1. / 1e-4 - 1.))
    for name, cfg in PASSES.items():
        tex = torch.ones(args.batch, f.shape[0], 1, 3, device=dev) if name == "mask" else torch.tensor(ftex, device=dev)
        tex.requires_grad_(True)
        kw = dict(image_size=args.size, dist_func="euclidean", aggr_func_alpha="prod", **cfg)
        img = srf.soft_rasterize(fv_t, tex, **kw)
        g = torch.randn_like(img)
        t_f = timeit(lambda: srf.soft_rasterize(fv_t, tex, **kw))
        t_fb = timeit(lambda: torch.autograd.grad(srf.soft_rasterize(fv_t, tex, **kw), (fv_t, tex), g))
        pairs = native.count_pairs(fv_t.detach().reshape(args.batch, -1, 9).contiguous(), args.size, cfg["sigma_val"], dist_eps)
        res[name] = {"fwd_ms": round(t_f, 4), "fwd_bwd_ms": round(t_fb, 4), "pairs_bbox": pairs,
                     "Gpairs_per_s_fwd": round(pairs / t_f / 1e6, 2)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
