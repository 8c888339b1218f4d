import os, sys, numpy as np, torch
sys.path[:0]=[os.getcwd(), os.getcwd()+"/self-corr-pose_amd", os.getcwd()+"/tests"]
import scenes
from oracle import ref_gpu
from test_softras_gpu import PASSES, hip_render
v, f = scenes.bottle_like(4)
fv, ftex = scenes.raster_inputs(v, f, 2, seed=77, tex="rand")
for pname in ("softtex", "depth"):
    kw = dict(image_size=512, dist_func="euclidean", aggr_func_alpha="prod", **PASSES[pname])
    got = hip_render(fv, ftex, None, **kw)
    ref = ref_gpu.render(fv, ftex, variant="nocontract", **kw)
    d = np.abs(got["soft_colors"].astype(np.float64) - ref["soft_colors"])
    bad = np.argwhere(d > 2e-6 + 1e-5*np.abs(ref["soft_colors"]))
    print(pname, "bad", len(bad))
    for b in bad[:12]:
        print("  ", tuple(b), got["soft_colors"][tuple(b)], ref["soft_colors"][tuple(b)])
