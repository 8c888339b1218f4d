"""Data-parallel path on CPU: 2 processes, gloo backend (the GPU box uses the same code over RCCL).

(1) FlatGradients: p.grad are views of one flat buffer; bucketed all-reduce launched from autograd hooks equals the
    mean of the per-rank gradients, parameters without a gradient are handled, replicas (and buffers) start identical
    after the broadcast; buckets go out BEFORE backward ends (overlap).
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
    from scp_amd.parallel import FlatGradients
    torch.manual_seed(rank)           # replicas start DIFFERENT on purpose
    model = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.Linear(33, 5), torch.nn.Linear(5, 3))
    model.register_buffer("running", torch.full((3,), float(rank)))
    red = FlatGradients(model.parameters(), bucket_bytes=600)     # forces several buckets
    assert len(red.buckets) > 1
    # the flat buffer is laid out in reverse registration order, buckets are contiguous slices that tile it
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == red.flat.numel()
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))
    assert red.span[id(list(model.parameters())[-1])][0] == 0
    red.broadcast_parameters(model, 0)
    w0 = [p.detach().clone() for p in model.parameters()] + [model.running.clone()]
    x = torch.randn(4, 7, generator=torch.Generator().manual_seed(100 + rank))
    # local gradients first (no hooks armed), for the expected value
    h = model[1](model[0](x))
    model[2](h).square().sum().backward()                  # model[3] gets no gradient at all
    local = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    for p in model.parameters():
        p.grad = None
    # step 1 learns that model[3] never produces a gradient (nothing can overlap yet: its bucket comes first);
    # step 2 no longer waits for it
    for step in range(2):
        red.prepare()
        assert all(p.grad.data_ptr() == red.views[id(p)].data_ptr() for p in model.parameters())
        h = model[1](model[0](x))
        loss = model[2](h).square().sum()
        loss.backward()
        launched_during = red.launched_in_backward
        flat = red.finish()
    assert flat.data_ptr() == red.flat.data_ptr()
    flat.div_(world)
    gathered = [None] * world
    dist.all_gather_object(gathered, [None if g is None else g.numpy() for g in local])
    for i, p in enumerate(model.parameters()):
        assert p.grad.data_ptr() == red.views[id(p)].data_ptr()          # still views of the flat buffer
        parts = [torch.zeros_like(p) if g[i] is None else torch.tensor(g[i]) for g in gathered]
        torch.testing.assert_close(p.grad, sum(parts) / world, rtol=1e-6, atol=1e-7)
    sync = [None] * world
    dist.all_gather_object(sync, [w.numpy() for w in w0])
    for a, b in zip(sync[0], sync[1]):
        assert (a == b).all()
    # checkpoint statistics: averaged into a COPY; the live buffers keep their per-rank values, non-statistics are not touched
    model.register_buffer("running_mean", torch.full((3,), float(rank)))
    avg = red.averaged_running_stats(model)
    assert set(avg) == {"running_mean"} and torch.equal(avg["running_mean"], torch.full((3,), 0.5))
    assert torch.equal(model.running_mean, torch.full((3,), float(rank))) and torch.equal(model.running, torch.full((3,), 0.0))
    out.put((rank, launched_during))
    dist.destroy_process_group()


def _late_parameter_worker(rank, world, port, out):
    """a parameter that had no gradient in the previous step starts producing one.
    (a) its gradient arrives BEFORE its bucket is complete: the bucket must not go out from a hook any more (finish()
        launches it after backward) and the averaged gradients must be exact;
    (b) it arrives AFTER the bucket was all-reduced: unrecoverable, must raise instead of training on diverged replicas."""
    _init(rank, world, port)
    from scp_amd.parallel import FlatGradients
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(6, 6), torch.nn.Linear(6, 6), torch.nn.Linear(6, 6)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    red = FlatGradients(params, bucket_bytes=1 << 20)          # one bucket
    assert len(red.buckets) == 1
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(7 + rank))
    red.prepare()
    b(x).square().sum().backward()                             # step 1: only b is used -> a, c are learned as silent
    red.finish(keep_unused_none=True)
    assert all(p.grad is None for p in list(a.parameters()) + list(c.parameters()))       # unused: AdamW would skip them
    assert all(p.grad is not None for p in b.parameters())
    # (a) c(b(x)): c's gradients come first, while b's are still pending
    red.prepare()
    assert all(p.grad is not None and p.grad.data_ptr() == red.views[id(p)].data_ptr() for p in params)
    c(b(x)).square().sum().backward()
    assert red.launched_in_backward == 0                       # the poisoned bucket waits for finish()
    flat = red.finish().clone().div_(world)
    local = torch.autograd.grad(c(b(x)).square().sum(), list(b.parameters()) + list(c.parameters()))
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.numpy() for g in local])
    for i, p in enumerate(list(b.parameters()) + list(c.parameters())):
        o, n = red.span[id(p)]
        want = sum(torch.tensor(g[i]) for g in gathered) / world
        torch.testing.assert_close(flat[o:o + n].view_as(p), want, rtol=1e-6, atol=1e-7)
    # (b) a step that uses b only (a and c silent again), then b(a(x)): b completes the bucket, a's gradient arrives after
    # the all-reduce was enqueued
    red.prepare(); b(x).square().sum().backward(); red.finish()
    red.prepare()
    raised = False
    try:
        b(a(x)).square().sum().backward()
    except RuntimeError as e:
        raised = "after its bucket" in str(e)
    # every rank raises at the same point, so no collective is left half-issued; reset_static_graph() is the documented way
    red._armed = False
    for w in red._work:
        if w is not None:
            w.wait()
    red.reset_static_graph()
    red.prepare()
    b(a(x)).square().sum().backward()
    red.finish()
    out.put((rank, raised))
    dist.destroy_process_group()


def _overlap_worker(rank, world, port, out):
    """the bucket holding the LAST layers' gradients must be on the wire before backward has reached the first layer:
    a backward hook on the first layer records how many buckets had been launched when autograd got there"""
    _init(rank, world, port)
    from scp_amd.parallel import FlatGradients
    torch.manual_seed(0)
    layers = [torch.nn.Linear(64, 64) for _ in range(8)]
    model = torch.nn.Sequential(*layers)
    red = FlatGradients(model.parameters(), bucket_bytes=2 * (64 * 64 + 64) * 4)     # 2 layers per bucket -> 4 buckets
    assert len(red.buckets) == 4
    seen = {}

    def at_first_layer(g):
        seen.setdefault("at_first_layer", red.launched_in_backward)
    layers[0].weight.register_hook(at_first_layer)
    red.prepare()
    x = torch.randn(16, 64, generator=torch.Generator().manual_seed(rank))
    model(x).square().mean().backward()
    seen["at_backward_end"] = red.launched_in_backward
    red.finish()
    out.put((rank, (seen["at_first_layer"], seen["at_backward_end"])))
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
    torch.manual_seed(rank)            # different seeds per rank ON PURPOSE: the init broadcast must make replicas identical
    tr = Trainer(opts, prior=scenes.bottle_like(2), device="cpu")
    assert tr.reducer is not None and tr.reducer.world == world       # Trainer.__init__ broadcast rank 0's replica
    data = synth.make_batch(1, 2, 128, seed=10 + rank, device="cpu")   # different data per rank
    total, aux, _ = tr.step(data)
    flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters() if p.requires_grad])
    both = [None] * world
    dist.all_gather_object(both, (float(total), flat.numpy()))
    assert both[0][0] != both[1][0]                    # ranks really saw different batches
    assert (both[0][1] == both[1][1]).all()            # ... and still hold identical replicas
    out.put((rank, "ok"))
    dist.destroy_process_group()


def _run(worker, world=2, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = dict(q.get(timeout=5) for _ in range(world))
    assert sorted(got) == list(range(world))
    return got


def test_flat_gradients_gloo_world2():
    got = _run(_reducer_worker)
    assert all(v >= 1 for v in got.values())           # at least one bucket went out from a hook, i.e. inside backward


def test_bucket_all_reduce_overlaps_backward_gloo_world2():
    got = _run(_overlap_worker)
    for first, end in got.values():
        assert first >= 2, "no bucket was launched before backward reached the first layer"
        assert end >= 3


def test_trainer_step_data_parallel_gloo_world2():
    _run(_trainer_worker)


def test_parameter_that_becomes_used_gloo_world2():
    got = _run(_late_parameter_worker)
    assert all(got.values()), "a gradient that arrived after its bucket's all-reduce must raise"


def _one_rank_uses_it_worker(rank, world, port, out):
    """a parameter that receives a gradient on rank 0 only: every replica must keep its (averaged) gradient -- not None on the rank
    that did not use it -- so that AdamW steps it identically everywhere; a parameter unused on BOTH ranks ends with grad None.
    Later: the used set changes on one rank without reset_static_graph() -> check_static_graph() raises on every rank."""
    _init(rank, world, port)
    from scp_amd.parallel import FlatGradients
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    red = FlatGradients(params)
    red.broadcast_parameters()
    x = torch.randn(4, 8, generator=torch.Generator().manual_seed(3 + rank))

    def step(use_b):
        red.prepare()
        y = a(x)
        if use_b:
            y = y + b(x)
        y.square().mean().backward()
        red.finish(keep_unused_none=True).div_(world)
    step(use_b=(rank == 0))
    ok = all(p.grad is not None for p in a.parameters()) and all(p.grad is not None for p in b.parameters())
    ok = ok and all(p.grad is None for p in c.parameters())
    gb = b.weight.grad.clone()
    both = [None] * world
    dist.all_gather_object(both, gb.numpy())
    ok = ok and bool((both[0] == both[1]).all()) and float(gb.abs().sum()) > 0      # the same averaged gradient on both ranks
    red.check_static_graph()                                                        # nothing has changed yet
    step(use_b=False)                                                               # rank 0 stops using b: the set changed
    raised = False
    try:
        red.check_static_graph()
    except RuntimeError:
        raised = True
    red.reset_static_graph()
    step(use_b=False)
    ok = ok and raised and all(p.grad is None for p in b.parameters())
    out.put((rank, ok))
    dist.destroy_process_group()


def test_parameter_used_on_one_rank_only_gloo_world2():
    got = _run(_one_rank_uses_it_worker)
    assert all(got.values())


def _sync_bn_worker(rank, world, port, out):
    """Trainer(sync_bn=True) (the reference's multi-GPU choice, trainer.py:67): two ranks with B images each normalise with the
    statistics of the JOINT 2B batch -- the encoder's features on each rank equal those of one process that sees all 2B images.
    torch's SyncBatchNorm only takes GPU tensors: two RCCL ranks on two GPUs."""
    for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = "cuda:%d" % rank
    import scenes
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.trainer import Trainer
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True, total_iters=50, img_size=64, corr_h=16, corr_w=16, ngpu=world)
    torch.manual_seed(0)
    tr = Trainer(opts, prior=scenes.bottle_like(2), device=dev, sync_bn=True)
    assert any(isinstance(m, torch.nn.SyncBatchNorm) for m in tr.model.modules())
    enc = tr.model.encoder
    enc.random_jitter = torch.nn.Identity()
    imgs = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(7)).to(dev)
    _, mine = enc.encode_img(imgs[2 * rank:2 * rank + 2])
    # the joint batch through plain BatchNorm in one process (same weights: built from the same seed, broadcast from rank 0)
    torch.manual_seed(0)
    ref = Trainer(opts, prior=scenes.bottle_like(2), device=dev, sync_bn=False, process_group=None)
    ref.model.load_state_dict(tr.model.state_dict(), strict=False)
    ref.model.encoder.random_jitter = torch.nn.Identity()
    _, joint = ref.model.encoder.encode_img(imgs)
    err = float((mine - joint[2 * rank:2 * rank + 2]).abs().max())
    out.put((rank, err))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sync_batchnorm_equals_joint_batch_rccl_world2():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (SyncBatchNorm takes GPU tensors only; RCCL refuses two ranks on one device)")
    got = _run(_sync_bn_worker)
    assert all(v <= 1e-5 for v in got.values()), got


def _rccl_two_rank_worker(rank, world, port, out):
    """2 real RCCL ranks on 2 GPUs: 1 x (2B) equals 2 x B (mean of the per-rank gradients == gradient of the mean loss over the
    joint batch), and >= 2 buckets are enqueued from inside backward"""
    for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from scp_amd import streams
    from scp_amd.parallel import FlatGradients
    streams.MODE = "overlap"                 # this test is about the buckets that go out from inside backward (scp_amd/streams.py)
    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(256, 256) for _ in range(6)]).cuda()
    red = FlatGradients(net.parameters(), bucket_bytes=2 * (256 * 256 + 256) * 4)
    red.broadcast_parameters(net, 0)
    xs = torch.randn(2 * 8, 256, generator=torch.Generator().manual_seed(5)).cuda()
    ref = torch.autograd.grad(net(xs).square().sum(1).mean(), list(net.parameters()))          # one process, 2B samples
    red.prepare()
    net(xs[rank * 8:(rank + 1) * 8]).square().sum(1).mean().backward()                        # this rank's B samples
    launched = red.launched_in_backward
    flat = red.finish()
    flat.div_(world)
    torch.cuda.synchronize()
    for p, g in zip(net.parameters(), ref):
        torch.testing.assert_close(p.grad, g, rtol=2e-5, atol=1e-6)
    out.put((rank, launched))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_rccl_gradient_equals_joint_batch():
    """needs >= 2 visible GPUs (the round-end 1-GPU box skips it; an 8-GPU node runs it)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: RCCL refuses two ranks on one device")
    got = _run(_rccl_two_rank_worker)
    assert all(v >= 2 for v in got.values()), got


def _rccl_single_rank_worker(rank, world, port, out, mode):
    for p in (ROOT, os.path.join(ROOT, "self-corr-pose_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from scp_amd import streams
    streams.MODE = mode
    from scp_amd.parallel import FlatGradients
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 8, 3, padding=1),
                                  torch.nn.Flatten(), torch.nn.Linear(8 * 16 * 16, 10)).cuda().to(memory_format=torch.channels_last)
        x = torch.randn(4, 3, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
        ref = torch.autograd.grad(net(x).square().sum(), list(net.parameters()))
        red = FlatGradients(net.parameters(), bucket_bytes=4096, distributed=True, force_collectives=True)
        assert red.active and (red.comm_stream is not None) == (mode == "overlap") and len(red.buckets) > 1
        for step in range(2):
            red.prepare()
            net(x).square().sum().backward()
            flat = red.finish()
            torch.cuda.synchronize()
            if mode == "overlap":
                assert red.launched_in_backward >= len(red.buckets) - 1, (red.launched_in_backward, len(red.buckets))
            else:
                assert red.launched_in_backward == 0
            for p, g in zip(net.parameters(), ref):
                assert p.grad.data_ptr() == red.views[id(p)].data_ptr() and p.grad.stride() == p.stride()
                torch.testing.assert_close(p.grad, g, rtol=1e-5, atol=1e-6)
            assert flat.data_ptr() == red.flat.data_ptr()
        out.put((rank, True))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["overlap", "serial"])
def test_flat_gradients_over_rccl_single_rank(mode):
    """the CUDA side of the all-reduce (views with channels_last strides, async RCCL work objects; SCP_STREAMS=overlap, the default:
    communication stream, events, buckets launched from the gradient hooks; serial: no communication stream, every bucket goes out from
    finish()) on ONE GPU: a 1-rank nccl group with force_collectives -- the sum over one rank must equal plain autograd and a second
    step must reuse the same buffers.  In its own process like the 2-rank tests: an RCCL group inside the long-lived pytest process
    aborted twice in a backward when this file ran after tests/test_step_gpu.py (round 5; not reproduced in isolation)."""
    _run(_rccl_single_rank_worker, world=1, extra=(mode,))
