"""tools/pipeline_probe.py -- does software-pipelining the frozen DINO ViT across iterations (Trainer.step(next_data=...): the
next batch's features are enqueued on the side stream before this step's backward) change the step time?  Two distinct
synthetic batches alternate, every step runs exactly one ViT pass either way."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
import bench  # noqa: E402
from scp_amd import synthetic  # noqa: E402

tr, opts = bench.build_trainer("cuda", 1)
a = synthetic.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100, device="cuda")
b = synthetic.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=101, device="cuda")
for _ in range(6):
    tr.step(a)


def run(pipelined, steps=30):
    batches = (a, b)
    for i in range(4):
        tr.step(batches[i % 2], next_data=batches[(i + 1) % 2] if pipelined else None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[i % 2], next_data=batches[(i + 1) % 2] if pipelined else None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    print("in-step ViT   : %.2f ms/step" % run(False))
    print("pipelined ViT : %.2f ms/step" % run(True))
