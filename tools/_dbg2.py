import os, sys, numpy as np, torch
sys.path[:0]=[os.getcwd(), os.getcwd()+"/self-corr-pose_amd", os.getcwd()+"/tests"]
import scenes
from oracle import ref_gpu
from test_softras_gpu import PASSES, hip_render
DEV="cuda"
if "dual" in sys.argv:
    from scp_amd.soft_renderer import functional as srf
    v, f = scenes.bottle_like(3)
    fv, ftex = scenes.raster_inputs(v, f, 5, seed=31, tex="depth")
    _, fcanon = scenes.raster_inputs(v, f, 5, seed=31, tex="canon")
    size=int(os.environ.get('SZ','256'))
    common = dict(image_size=size, dist_func="euclidean", aggr_func_alpha="prod")
    fv_a = torch.tensor(fv, device=DEV); tex_a = torch.tensor(ftex, device=DEV); canon = torch.tensor(fcanon, device=DEV)
    depth = srf.soft_rasterize(fv_a, tex_a, **common, **PASSES["depth"])
    hard = srf.soft_rasterize(fv_a, canon, **common, **PASSES["hardtex"])
    depth2, hard2 = srf.soft_rasterize_dual(fv_a, tex_a, canon, size, PASSES["depth"]["background_color"], PASSES["hardtex"]["background_color"], sigma_val=1e-4, gamma_val=1e-4)
    print("depth equal", torch.equal(depth, depth2), (depth-depth2).abs().max().item(), "hard equal", torch.equal(hard, hard2), (hard-hard2).abs().max().item())
    for ch in range(4):
        print(ch, (hard[:,ch]!=hard2[:,ch]).sum().item(), (depth[:,ch]!=depth2[:,ch]).sum().item())
else:
    pname="softtex"
    rng = np.random.default_rng(7)
    n_f, size = 1500, 160
    a = rng.uniform(-1.1, 1.1, (2, n_f, 2)); ang = rng.uniform(0, 2 * np.pi, (2, n_f))
    length = 10 ** rng.uniform(-2.2, -0.3, (2, n_f)); height = length * 10 ** rng.uniform(-3.5, 0, (2, n_f))
    d = np.stack((np.cos(ang), np.sin(ang)), -1); nrm = np.stack((-np.sin(ang), np.cos(ang)), -1)
    b = a + d * length[..., None]
    c = a + d * (length * rng.uniform(-0.2, 1.2, (2, n_f)))[..., None] + nrm * height[..., None] * rng.choice([-1, 1], (2, n_f))[..., None]
    tri = np.stack((a, b, c), 2); z = rng.uniform(3.0, 9.0, (2, n_f, 3, 1))
    fv = np.concatenate((tri, z), -1).astype(np.float32)
    fv[0, 10] = fv[0, 10, :1]; fv[0, 11, 2] = fv[0, 11, 1]; fv[1, 12, 2, :2] = 0.5 * (fv[1, 12, 0, :2] + fv[1, 12, 1, :2])
    ftex = rng.uniform(0, 1, (2, n_f, 3, 3)).astype(np.float32)
    kw = dict(image_size=size, dist_func="euclidean", aggr_func_alpha="prod", **PASSES[pname])
    got = hip_render(fv, ftex, None, **kw)
    ref = ref_gpu.render(fv, ftex, variant="nocontract", **kw)
    dd = np.abs(got["soft_colors"].astype(np.float64) - ref["soft_colors"])
    bad = np.argwhere(dd > 2e-6 + 1e-5*np.abs(ref["soft_colors"]))
    print("bad", len(bad))
    for bb in bad[:10]:
        print("  ", tuple(int(x) for x in bb), got["soft_colors"][tuple(bb)], ref["soft_colors"][tuple(bb)], "aggr", got["aggrs_info"][bb[0],:,bb[2],bb[3]], ref["aggrs_info"][bb[0],:,bb[2],bb[3]])
