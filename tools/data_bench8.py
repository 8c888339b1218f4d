"""tools/data_bench8.py -- the input pipeline under EIGHT ranks on one host (8 x 32 images x 30 it/s = 7 680 images/s for the
8-GPU run, 960 per rank): 8 processes, each `data_loader(opts)` with local_rank = r, ngpu = 8 and W decode workers, all staging to the one
GPU of this box (crop / resize on the device, csrc/crop_resize.hip).  Synthetic on-disk set in the Wild6D layout (tests/wild6d_synth.py:
640 x 480 JPEG + mask PNG + depth PNG per frame).  Prints per-rank and aggregate images/s as one JSON line."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = int(os.environ.get("WORKERS", "12"))
RANKS = int(os.environ.get("RANKS", "8"))
CHILD = r'''
import json, os, sys, time, types
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "self-corr-pose_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
from scp_amd.data import data_loader
rank = %(rank)d
opts = types.SimpleNamespace(train_list=%(train_list)r, dataset_path=%(data)r, batch_size=8, repeat=4, ngpu=%(ranks)d, total_iters=40, img_size=256,
                             no_stretch=False, use_depth=True, local_rank=rank, num_workers=%(workers)d)
np.random.seed(rank)
loader, _ = data_loader(opts)
it = iter(loader)
next(it); next(it)
torch.cuda.synchronize()
t = time.perf_counter()
cnt = 0
for b in it:
    cnt += b["img"].shape[0]
torch.cuda.synchronize()
print(json.dumps({"rank": rank, "images": cnt, "fps": cnt / (time.perf_counter() - t)}))
'''
if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import wild6d_synth
    data = os.path.join(tempfile.mkdtemp(prefix="scp_wild6d_bench8_"), "wild6d")
    train_list = wild6d_synth.write_dataset(data, n_obj=4, n_seq=4, n_frames=24, w=640, h=480, seed=1)
    procs = [subprocess.Popen([sys.executable, "-c", CHILD % dict(root=ROOT, rank=r, train_list=train_list, data=data, ranks=RANKS, workers=W)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(RANKS)]
    rows = []
    for p in procs:
        out, err = p.communicate()
        line = [l for l in out.splitlines() if l.startswith("{")]
        rows.append(json.loads(line[-1]) if line else {"error": err.strip().splitlines()[-1][:200] if err.strip() else "no output"})
    fps = [r.get("fps", 0.0) for r in rows]
    print(json.dumps({"what": "%d concurrent ranks x data_loader (batch 32 = 8 videos x 4 frames, 640x480 -> 256x256, device crop/resize), %d decode "
                              "workers per rank, one host" % (RANKS, W), "per_rank_fps": [round(f, 1) for f in fps], "aggregate_fps": round(sum(fps), 1),
                      "needed_per_rank_at_30_it_s": 960, "cores": os.cpu_count(), "rows": rows}))
