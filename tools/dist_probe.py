"""tools/dist_probe.py -- per-step wall time of the N-rank launch path on ONE device (all ranks on cuda:0, gloo):
    SCP_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_probe.py
gloo's CUDA all-reduce is erratic here (0.1-20 s per step); it only
exercises the code path, the real multi-GPU run goes over RCCL."""
import os, sys, time, cProfile, pstats
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench, synth
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("SCP_DIST_BACKEND", "gloo"), init_method="env://")
tr, opts = bench.build_trainer("cuda:0", world)
data = synth.make_batch(opts.batch_size, opts.repeat, opts.img_size, seed=100 + rank, device="cuda:0")
for i in range(6):
    t = time.perf_counter(); tr.step(data); torch.cuda.synchronize()
    if rank == 0: print("step %d: %.1f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
if rank == 0:
    pr = cProfile.Profile(); pr.enable()
tr.step(data); torch.cuda.synchronize()
if rank == 0:
    pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
dist.barrier(); dist.destroy_process_group()
