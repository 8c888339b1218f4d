"""tools/render_profile.py -- Renderer.render_all + the image-space losses, forward + backward, in
isolation at the bench size: wall time and the per-op GPU time table."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import synth  # noqa: E402
from scp_amd import losses  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
m = tr.model
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
img, mask, depth, _, _, _, _, foc_crop, _, pp_crop, _, _ = data
B = img.shape[0]
with torch.no_grad():
    mean_v = m.mesh.mean_v[None].expand(B, -1, -1)
    _, _, pred_v0, rot0, trans0, scale = m.encoder(img, mean_v, pp_crop, foc_crop)
faces = m.mesh.faces[None].expand(B, -1, -1)
tex0 = torch.rand(B, m.mesh.num_verts, 3, device="cuda")


def run():
    pred_v, rot, trans, tex = (t.detach().clone().requires_grad_(True) for t in (pred_v0, rot0, trans0, tex0))
    (mask_r, tex_r, depth_r, match_gt, imatch_gt, tex_mask, depth_mask, match_mask, dw) = m.renderer.render_all(
        pred_v, faces, tex, foc_crop, pp_crop, rot, trans, scale)
    loss = (losses.compute_mask_loss(img, mask, mask_r).mean() + losses.compute_texture_loss(img, mask, tex_r, tex_mask).mean() +
            losses.compute_depth_loss(depth, depth_r, depth_mask, mask)[0].mean() + imatch_gt.square().mean())
    loss.backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    run()
torch.cuda.synchronize()
print("render_all + image losses, fwd+bwd: %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)[:40]
tot = sum(r.self_device_time_total for r in prof.key_averages())
print("total GPU time %.2f ms in %d kernels" % (tot / 1e3, sum(r.count for r in prof.key_averages() if r.self_device_time_total > 0)))
for r in rows:
    print("%8.3f ms x%-4d %s" % (r.self_device_time_total / 1e3, r.count, r.key[:100]))
