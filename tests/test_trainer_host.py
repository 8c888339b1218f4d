"""Host logic of the training step (scp_amd/trainer.py, optimizers.py, flags.py) on CPU."""
import math

import pytest
import torch

import scenes


@pytest.fixture(scope="module")
def trainer():
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.trainer import Trainer
    dino.ALLOW_RANDOM_INIT = True
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True, total_iters=50, img_size=128, corr_h=32,
                   corr_w=32, pretrain_k=40)
    torch.manual_seed(0)
    return Trainer(opts, prior=scenes.bottle_like(2), device="cpu")


def _fill(tr, fn):
    for i, p in enumerate(tr._trainable):
        p.grad = fn(i, p)


def test_collect_grad_clips_per_group_like_the_reference(trainer):
    """trainer.py:132-150: mean_v clipped to 1, shapenerf params to 1 (jointly), pose_predictor to 0.1"""
    tr = trainer
    _fill(tr, lambda i, p: torch.full_like(p, 0.5))
    expect = {}
    for name, group, mx in (("mean_v", tr._mean_v, 1.), ("shapenerf", tr._shapenerf, 1.), ("pose", tr._pose, 0.1)):
        total = math.sqrt(sum(0.25 * p.numel() for p in group))
        expect[name] = (total, min(1.0, mx / (total + 1e-6)))
    norms = tr.collect_grad()
    for (name, (total, coef)), got in zip(expect.items(), norms):
        assert abs(float(got) - total) <= 1e-4 * total
    for group, key in ((tr._mean_v, "mean_v"), (tr._shapenerf, "shapenerf"), (tr._pose, "pose")):
        for p in group:
            torch.testing.assert_close(p.grad, torch.full_like(p, 0.5 * expect[key][1]), rtol=1e-5, atol=0)
    other = [p for p in tr._trainable if all(p is not q for g in (tr._mean_v, tr._shapenerf, tr._pose) for q in g)]
    assert other and all(torch.equal(p.grad, torch.full_like(p, 0.5)) for p in other)   # backbone/featnet untouched


def test_collect_grad_drops_the_step_on_nan(trainer):
    tr = trainer
    _fill(tr, lambda i, p: torch.ones_like(p))
    tr._trainable[7].grad.view(-1)[0] = float("nan")
    tr._trainable[9].grad.view(-1)[0] = float("inf")
    tr.collect_grad()
    assert all((p.grad == 0).all() for p in tr._trainable)


def test_optimizer_groups_and_onecycle(trainer):
    """optimizers.py:16-73: five name-based groups with vert/cam lr ratios, OneCycle (pct 0.05, div 25)"""
    tr = trainer
    groups = tr.optim.optimizer.param_groups
    assert len(groups) == 5 and len(groups[0]["params"]) == 1            # mean_v
    lr = tr.opts.learning_rate
    for g, ratio in zip(groups, (tr.opts.vert_lr_ratio, tr.opts.cam_lr_ratio, 1, 1, 1)):
        assert abs(g["initial_lr"] - ratio * lr / 25) < 1e-12 and g["weight_decay"] == 1e-4
    n_opt = sum(len(g["params"]) for g in groups)
    # frozen DINO is left out; mesh.faces / mesh.symm_rots match no group (the reference prints
    # "*found unknown params" for them, optimizers.py:35)
    n_named = sum(1 for n, p in tr.model.named_parameters()
                  if "pretrain_corr_net" not in n and n not in ("mesh.faces", "mesh.symm_rots"))
    assert n_opt == n_named


def test_flagfile_parser(tmp_path):
    from scp_amd.flags import PRESETS, load_flagfile
    text = "--category=laptop\n--batch_size=8\n--repeat=4\n--depth_offset=5\n--use_depth=True\n" \
           "--rotation_offset=0.2,0.0,0.0,0.0,-0.2,0.2\n--divide_fn=both\n--pretrain_k=200\n--symmetry_idx=1\n"
    f = tmp_path / "base_config.txt"
    f.write_text(text)
    o = load_flagfile(str(f), train=True)
    assert o.batch_size == 8 and o.repeat == 4 and o.depth_offset == 5.0 and o.use_depth is True
    assert o.rotation_offset == [0.2, 0.0, 0.0, 0.0, -0.2, 0.2] and o.divide_fn == "both" and o.train
    assert PRESETS["laptop_wild6d"]["pretrain_k"] == o.pretrain_k
    with pytest.raises(AttributeError):
        (tmp_path / "bad.txt").write_text("--no_such_flag=1\n")
        load_flagfile(str(tmp_path / "bad.txt"))


def test_category_presets_equal_the_shipped_flag_sets():
    """the five Wild6D presets restate config/<category>_wild6d/base_config.txt (values recorded here so the check runs
    without the reference): the fields that differ between categories"""
    from scp_amd.flags import Options
    expect = {
        "bottle": dict(symmetry_idx=0, cycle_loss_wt=0.02, cycle_loss_pretrain_wt=0.05, vert_lr_ratio=0.1, cam_lr_ratio=0.1,
                       rotation_offset=[0.1, 0.0, 0.0, 0.0, 0.1, -0.1], base_rot=[1, 0, 0, 0, 1, 0, 0, 0, 1]),
        "bowl": dict(symmetry_idx=0, cycle_loss_wt=0.02, cycle_loss_pretrain_wt=0.005, vert_lr_ratio=0.1, cam_lr_ratio=0.2,
                     rotation_offset=[0.2, 0.0, 0.0, 0.0, -0.2, 0.2], base_rot=[1, 0, 0, 0, -1, 0, 0, 0, 1]),
        "camera": dict(symmetry_idx=-1, cycle_loss_wt=0.02, cycle_loss_pretrain_wt=0.005, vert_lr_ratio=0.1, cam_lr_ratio=0.1,
                       rotation_offset=[0.2, 0.0, -0.1, 0.0, -0.2, 0.2], base_rot=[1, 0, 0, 0, -1, 0, 0, 0, 1]),
        "laptop": dict(symmetry_idx=1, cycle_loss_wt=0.01, cycle_loss_pretrain_wt=0.02, vert_lr_ratio=0.01, cam_lr_ratio=0.1,
                       rotation_offset=[0.2, 0.0, 0.0, 0.0, -0.2, 0.2], base_rot=[0, 0, 1, 0, -1, 0, -1, 0, 0]),
        "mug": dict(symmetry_idx=1, cycle_loss_wt=0.01, cycle_loss_pretrain_wt=0.02, vert_lr_ratio=0.01, cam_lr_ratio=0.1,
                    rotation_offset=[0.1, 0.0, 0.0, 0.0, -0.1, 0.1], base_rot=[1, 0, 0, 0, -1, 0, 0, 0, 1]),
    }
    for cat, fields in expect.items():
        o = Options(cat + "_wild6d")
        assert o.category == cat and o.batch_size == 8 and o.repeat == 4 and o.pretrain_k == 200 and o.divide_fn == "both"
        for k, v in fields.items():
            assert getattr(o, k) == v, (cat, k)


def test_resnet18_imagenet_weights_are_required_or_loaded(tmp_path):
    """the backbone must start from ImageNet weights like the reference (image_encoder.py:122); a missing
    file raises instead of silently falling back to a random initialisation"""
    import scp_amd.dino as dino
    from scp_amd import nets
    saved = dino.ALLOW_RANDOM_INIT
    try:
        dino.ALLOW_RANDOM_INIT = False
        with pytest.raises(FileNotFoundError):
            nets.ResNet_Encoder(str(tmp_path / "nope.pth"))
        nets.ResNet_Encoder(str(tmp_path / "nope.pth"), will_load_checkpoint=True)      # a checkpoint will overwrite it
        # a torchvision-style state_dict (same key names + fc.*) is loaded, fc dropped
        torch.manual_seed(1)
        donor = nets.ResNet18Trunk()
        sd = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in donor.state_dict().items()}
        sd["fc.weight"], sd["fc.bias"] = torch.zeros(1000, 512), torch.zeros(1000)
        torch.save(sd, tmp_path / "resnet18.pth")
        enc = nets.ResNet_Encoder(str(tmp_path / "resnet18.pth"))
        for k, v in enc.resnet.state_dict().items():
            assert torch.equal(v, sd[k]), k
        torch.save({"conv1.weight": sd["conv1.weight"]}, tmp_path / "bad.pth")
        with pytest.raises(RuntimeError):
            nets.ResNet_Encoder(str(tmp_path / "bad.pth"))
    finally:
        dino.ALLOW_RANDOM_INIT = saved


def test_flat_gradient_views_follow_the_parameter_layout():
    """channels_last convolution weights get channels_last gradient views (fused AdamW requires equal layouts), all views
    tile one flat buffer, and backward accumulates into them in place"""
    from scp_amd.parallel import FlatGradients
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1)).to(memory_format=torch.channels_last)
    fg = FlatGradients(net.parameters(), distributed=False)
    fg.prepare()
    for p in net.parameters():
        assert p.grad.stride() == p.stride() and p.grad.shape == p.shape
    x = torch.randn(2, 3, 5, 5)
    net(x).square().sum().backward()
    expect = torch.autograd.grad(net(x).square().sum(), list(net.parameters()))
    flat = fg.finish()
    assert flat.data_ptr() == fg.flat.data_ptr()
    for p, e in zip(net.parameters(), expect):
        assert p.grad.data_ptr() == fg.views[id(p)].data_ptr()
        torch.testing.assert_close(p.grad, e)
    assert abs(float(flat.abs().sum()) - float(sum(e.abs().sum() for e in expect))) < 1e-3 * float(flat.abs().sum())


def test_half_resolution_features_are_the_even_pixels_of_the_full_map():
    """ResNet_Decoder.forward(half_res=True) / Encoder.encode_img(half_res=True) (used for the rotated images of the
    rotation-cycle loss) == the full feature map at its even pixels, values and gradients"""
    import torch
    import scp_amd.dino as dino
    from scp_amd.encoder import Encoder
    from scp_amd.flags import Options
    from scp_amd.nets import ResNet_Decoder
    torch.manual_seed(0)
    for downsample in (4, 8):
        dec = ResNet_Decoder(is_proj=True, out_channel=16, downsample=downsample).double()
        pyr = [torch.randn(2, ch, s, s, dtype=torch.float64, requires_grad=True) for ch, s in ((64, 16), (128, 8), (256, 4), (512, 2))]
        full = dec(*pyr)[:, :, ::2, ::2]
        half = dec(*pyr, half_res=True)
        torch.testing.assert_close(half, full, rtol=1e-12, atol=1e-13)
        w = torch.randn_like(half)
        leaves = pyr + list(dec.parameters())
        g_half = torch.autograd.grad((half * w).sum(), leaves, allow_unused=True)
        g_full = torch.autograd.grad((full * w).sum(), leaves, allow_unused=True)
        for a, b in zip(g_half, g_full):
            assert (a is None) == (b is None)
            if a is not None:
                torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-12)
    # through the whole encoder (its output is cast to float32, hence the float32-level tolerance)
    opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True, img_size=64)
    old, dino.ALLOW_RANDOM_INIT = dino.ALLOW_RANDOM_INIT, True
    try:
        enc = Encoder(opts).eval()               # eval: BatchNorm on running statistics
    finally:
        dino.ALLOW_RANDOM_INIT = old
    enc.random_jitter = torch.nn.Identity()      # no colour-jitter draw between the two calls
    img = torch.rand(2, 3, 64, 64)
    _, full = enc.encode_img(img)
    _, half = enc.encode_img(img, half_res=True)
    c = opts.n_corr_feat
    hf = int(round(full.shape[-1] ** 0.5))
    torch.testing.assert_close(half, full.reshape(2, c, hf, hf)[:, :, ::2, ::2].reshape(2, c, -1), rtol=1e-5, atol=1e-6)


def test_sync_batchnorm_modules_take_the_stock_path():
    """Trainer(sync_bn=True) (the reference's multi-GPU choice, trainer.py:67) converts the trunk's BatchNorms to torch's
    SyncBatchNorm; the fused convolution + BatchNorm op must then decline (it implements per-rank statistics only) instead of
    silently normalising with local statistics"""
    import torch
    from scp_amd import fused_conv
    conv = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False)
    bn = torch.nn.SyncBatchNorm(64)
    calls = []
    orig = fused_conv._ConvBNAct.apply
    try:
        fused_conv._ConvBNAct.apply = staticmethod(lambda *a, **k: calls.append(1) or orig(*a, **k))
        bn.eval()                     # SyncBatchNorm's training forward needs a GPU process group; the dispatch is what is checked
        y = fused_conv.conv_bn_act(torch.randn(2, 64, 8, 8), conv, bn, relu=True)
    finally:
        fused_conv._ConvBNAct.apply = orig
    assert not calls and y.shape == (2, 64, 8, 8)


def test_flat_adamw_class_rows_are_stable_and_recycled():
    """scp_amd.optimizers.assign_classes (the host half of FlatAdamW's kernel-argument table): rows never move while there is room, a
    full table drops the classes nobody is in and renumbers, more live classes than rows is an error"""
    from scp_amd.optimizers import assign_classes
    ids = assign_classes({}, [(0, 0), (1, 0)], 4)
    assert ids == {(0, 0): 0, (1, 0): 1}
    assert assign_classes(ids, [(0, 0)], 4) is ids                              # nothing new: the same object, nothing to upload
    ids2 = assign_classes(ids, [(0, 0), (1, 3)], 4)
    assert ids2 == {(0, 0): 0, (1, 0): 1, (1, 3): 2} and ids == {(0, 0): 0, (1, 0): 1}
    ids3 = assign_classes(ids2, [(0, 0), (1, 4)], 4)
    assert ids3[(0, 0)] == 0 and ids3[(1, 4)] == 3
    ids4 = assign_classes(ids3, [(0, 0), (1, 5)], 4)                              # full: dead classes go, live ones are renumbered
    assert ids4 == {(0, 0): 0, (1, 5): 1}
    with pytest.raises(RuntimeError, match="classes"):
        assign_classes(ids4, [(g, 0) for g in range(5)], 4)


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (torch.distributed.run on
    127.0.0.1, a free port, one process per GPU: what scripts/train.sh:5-7 / train.py:29-36 do) and passes their status through; inside
    a rank (WORLD_SIZE set) it does not launch again"""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("scp_bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index(os.path.join(root, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a rank of that launch (WORLD_SIZE present) must go on to run, not launch again: without a GPU that is the "needs an MI355X" exit
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit) as e:
            bench.main()
        assert "MI355X" in str(e.value.code) and not seen


def test_wait_bounded_raises_device_stall_naming_the_busy_streams(monkeypatch):
    """host logic of scp_amd.streams.wait_bounded (the loops' only host<->device waits go through it): returns when the event completes,
    raises DeviceStall past the bound with the streams that still have work and the breadcrumb summary"""
    import time
    from scp_amd import streams

    class FakeEvent:
        def __init__(self, done_after):
            self.t0, self.done_after = time.monotonic(), done_after

        def query(self):
            return time.monotonic() - self.t0 >= self.done_after

    class FakeStream:
        def __init__(self, idle, handle=0):
            self.idle, self.cuda_stream = idle, handle

        def query(self):
            return self.idle

    streams.wait_bounded(FakeEvent(0.02), "quick", {"main": FakeStream(True)}, timeout=5.0)          # completes: no exception
    named = {"main": FakeStream(False, 1), "frozen-ViT side stream": FakeStream(True, 2), "rotation-cycle side stream": FakeStream(False, 3)}
    with pytest.raises(streams.DeviceStall) as e:
        streams.wait_bounded(FakeEvent(1e9), "Trainer.train: losses of iterations 1..2", lambda: named, timeout=0.05)
    msg = str(e.value)
    assert "iterations 1..2" in msg and "main" in msg and "rotation-cycle side stream" in msg and "frozen-ViT" not in msg
    # breadcrumbs: per stream, the last crumb reached and the first not reached
    monkeypatch.setattr(streams, "_crumbs", [("step 0: start", 1, FakeEvent(0)), ("step 0: forward done", 1, FakeEvent(1e9)),
                                             ("ViT prefetch: start", 2, FakeEvent(0))])
    rep = streams.crumb_report(named)
    assert "main: reached step 0: start, NOT reached step 0: forward done" in rep and "frozen-ViT side stream: all 1 crumbs reached" in rep
    assert streams.stall_timeout() == 300.0
    monkeypatch.setenv("SCP_DEVICE_TIMEOUT_S", "12")
    assert streams.stall_timeout() == 12.0
