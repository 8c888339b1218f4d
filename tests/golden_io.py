"""tests/golden_io.py -- loads the committed golden vectors (tests/golden/*.npz)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def softras_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "softras_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if "grad_soft_colors_q8" in d:
        d["grad_soft_colors"] = d.pop("grad_soft_colors_q8").astype(np.float32) / 8
    return d


def softras_kwargs(d):
    """keyword arguments of soft_rasterize() recorded with a SoftRas golden case"""
    return dict(image_size=int(d["image_size"]), background_color=[float(x) for x in d["background_color"]],
                near=1., far=100., fill_back=True, eps=1e-3, sigma_val=float(d["sigma_val"]),
                dist_func=str(d["dist_func"]), dist_eps=1e-4, gamma_val=float(d["gamma_val"]),
                aggr_func_rgb=str(d["aggr_func_rgb"]), aggr_func_alpha=str(d["aggr_func_alpha"]),
                texture_type=str(d["texture_type"]))
