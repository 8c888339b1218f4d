"""tools/data_bench.py -- input pipeline throughput (frames/s), 640x480 frames -> 256x256 batches of 32:
  host    : the reference-style item (decode + float64 resized_crop on the CPU), one process
  decode  : raw_item only (decode + box cut), one process
  device  : GpuCollator on pre-decoded raw items (staging fill + H2D + HIP crop/resize), per batch
  loader  : data_loader end to end with W workers"""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import wild6d_synth  # noqa: E402
from scp_amd.data import GpuCollator, Wild6DDataset, data_loader  # noqa: E402

W = int(os.environ.get("WORKERS", "16"))
root = os.path.join(tempfile.mkdtemp(prefix="scp_wild6d_bench_"), "wild6d")
train_list = wild6d_synth.write_dataset(root, n_obj=2, n_seq=2, n_frames=16, w=640, h=480, seed=1)
opts = types.SimpleNamespace(train_list=train_list, dataset_path=root, batch_size=8, repeat=4, ngpu=1, total_iters=24, img_size=256,
                             no_stretch=False, use_depth=True, local_rank=-1, num_workers=W)
np.random.seed(0)
ds = Wild6DDataset(opts)
n = 64
t = time.perf_counter(); [ds[i] for i in range(n)]; host = n / (time.perf_counter() - t)
t = time.perf_counter(); raws = [ds.raw_item(i) for i in range(n)]; decode = n / (time.perf_counter() - t)
col = GpuCollator(256, "cuda", True)
for _ in range(3):
    col(raws[:32])
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    col(raws[:32])
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t) / 20 * 1e3
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
loader, _ = data_loader(opts)
it = iter(loader)
next(it)
torch.cuda.synchronize()
t = time.perf_counter()
cnt = 0
for b in it:
    cnt += b["img"].shape[0]
torch.cuda.synchronize()
loader_fps = cnt / (time.perf_counter() - t)
print(json.dumps({"frame": "640x480 -> 256x256", "host_item_fps_1proc": host, "decode_cut_fps_1proc": decode,
                  "device_collate_ms_per_32": dev_ms, "device_collate_fps": 32 / dev_ms * 1e3,
                  "loader_fps": loader_fps, "workers": W, "cores": os.cpu_count()}))
