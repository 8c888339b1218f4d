import os, sys, time, torch
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench
from scp_amd import synthetic as synth
tr, opts = bench.build_trainer("cuda:0", 1)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda:0")
for _ in range(6): tr.step(data)
def timed(nd, n=20):
    for _ in range(3): tr.step(data, next_data=nd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step(data, next_data=nd)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print("no look-ahead   %.2f ms/step" % timed(None), flush=True)
    print("ViT look-ahead  %.2f ms/step" % timed(data), flush=True)
