"""Data-parallel path on CPU: 2 processes, gloo backend (the GPU box uses the same code over RCCL).

(1) GradientAllReducer: bucketing + averaging equals the mean of the per-rank gradients, parameters
    without a gradient on one rank are handled, replicas start identical after broadcast.
(2) Trainer.step with world_size 2: both ranks end the step with bit-identical parameters although
    they saw different batches (gradients were averaged before clipping / AdamW)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _reducer_worker(rank, world, port, out):
    _init(rank, world, port)
    from scp_amd.parallel import GradientAllReducer
    torch.manual_seed(rank)           # replicas start DIFFERENT on purpose
    model = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.Linear(33, 5), torch.nn.Linear(5, 3))
    red = GradientAllReducer(model, bucket_bytes=600)     # forces several buckets
    assert len(red.buckets) > 1
    red.broadcast_parameters(0)
    w0 = [p.detach().clone() for p in model.parameters()]
    x = torch.randn(4, 7, generator=torch.Generator().manual_seed(100 + rank))
    h = model[1](model[0](x))
    loss = model[2](h).square().sum()                      # model[3] gets no gradient at all
    loss.backward()
    local = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    red.all_reduce()
    gathered = [None] * world
    dist.all_gather_object(gathered, [None if g is None else g.numpy() for g in local])
    for i, p in enumerate(model.parameters()):
        parts = [torch.zeros_like(p) if g[i] is None else torch.tensor(g[i]) for g in gathered]
        torch.testing.assert_close(p.grad, sum(parts) / world, rtol=1e-6, atol=1e-7)
    sync = [None] * world
    dist.all_gather_object(sync, [w.numpy() for w in w0])
    for a, b in zip(sync[0], sync[1]):
        assert (a == b).all()
    out.put((rank, "ok"))
    dist.destroy_process_group()


def _trainer_worker(rank, world, port, out):
    _init(rank, world, port)
    import oracle_backend
    import scenes
    import synth
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.trainer import Trainer

    class MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    oracle_backend.install(MP())       # CPU stand-in for the HIP rasteriser (test only)
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True, total_iters=50, img_size=128, corr_h=32,
                   corr_w=32, pretrain_k=40, ngpu=world)
    torch.manual_seed(0)
    tr = Trainer(opts, prior=scenes.bottle_like(2), device="cpu")
    assert tr.reducer is not None
    tr.reducer.broadcast_parameters(0)
    data = synth.make_batch(1, 2, 128, seed=10 + rank, device="cpu")   # different data per rank
    total, aux, _ = tr.step(data)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters() if p.requires_grad])
    both = [None] * world
    dist.all_gather_object(both, (float(total), flat.numpy()))
    assert both[0][0] != both[1][0]                    # ranks really saw different batches
    assert (both[0][1] == both[1][1]).all()            # ... and still hold identical replicas
    out.put((rank, "ok"))
    dist.destroy_process_group()


def _run(worker, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == list(range(world))


def test_gradient_all_reducer_gloo_world2():
    _run(_reducer_worker)


def test_trainer_step_data_parallel_gloo_world2():
    _run(_trainer_worker)
