"""training input pipeline: host mirror vs vectors recorded from the reference's Wild6DDataset, device
crop+resize vs both."""
import os
import types

import numpy as np
import pytest
import torch

import wild6d_synth

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wild6d_items.npz"))


def _opts(tmp_path, **over):
    root = os.path.join(str(tmp_path), "wild6d")
    train_list = wild6d_synth.write_dataset(root, seed=0)
    o = dict(train_list=train_list, dataset_path=root, batch_size=2, repeat=3, ngpu=1, total_iters=4, img_size=64,
             no_stretch=False, use_depth=True, local_rank=-1, num_workers=0)
    o.update(over)
    return types.SimpleNamespace(**o)


def _dataset(tmp_path, **over):
    from scp_amd.data import Wild6DDataset
    np.random.seed(11)                               # the fixture run seeded the sampler like this
    return Wild6DDataset(_opts(tmp_path, **over))


SCALARS = ("center", "length", "foc", "foc_crop", "pp", "pp_crop", "idx", "frame_idx")


def test_sampler_and_items_match_reference(tmp_path):
    ds = _dataset(tmp_path)
    assert len(ds) == int(GOLD["n_items"])
    assert np.array_equal(np.array(ds.sample_list, dtype=np.int64), GOLD["sample_list"])
    np.random.seed(12)
    for i in range(len(ds)):
        e = ds[i]
        for k in SCALARS:
            ref = GOLD["i%02d_%s" % (i, k)]
            assert np.allclose(e[k].numpy().astype(np.float64), ref.astype(np.float64), rtol=1e-6, atol=0), (i, k)
        assert str(e["img"].dtype) == str(GOLD["i%02d_img_dtype" % i])
        if i < 10:
            assert np.abs(e["img"].numpy() - GOLD["i%02d_img" % i]).max() < 1e-6
            assert np.array_equal(e["mask"].numpy(), GOLD["i%02d_mask" % i])
            assert np.array_equal(e["depth"].numpy(), GOLD["i%02d_depth" % i])


def test_raw_items_describe_the_same_box(tmp_path):
    ds = _dataset(tmp_path)
    np.random.seed(12)
    raws = [ds.raw_item(i) for i in range(len(ds))]
    padded = 0
    for i, r in enumerate(raws):
        ih, iw, pt, pl, vh, vw = r["_crop"]["geom"]
        length = GOLD["i%02d_length" % i]
        assert (vh, vw) == (2 * int(length[1]), 2 * int(length[0]))
        assert r["_crop"]["img"].shape == (ih, iw, 3) and r["_crop"]["depth"].dtype == np.uint16
        assert 0 <= pt and 0 <= pl and pt + ih <= vh and pl + iw <= vw
        padded += (ih, iw) != (vh, vw)
    assert padded > 0, "the synthetic set is built so that some boxes leave the frame"


@pytest.mark.gpu
def test_device_batches_match_reference(tmp_path):
    from scp_amd.data import GpuCollator
    ds = _dataset(tmp_path)
    np.random.seed(12)
    raws = [ds.raw_item(i) for i in range(len(ds))]
    np.random.seed(12)
    cpu = [ds[i] for i in range(len(ds))]
    batch = GpuCollator(64, "cuda", True)(raws)
    img, mask, depth = batch["img"].cpu(), batch["mask"].cpu(), batch["depth"].cpu()
    assert img.dtype == torch.float32 and img.shape == (len(ds), 3, 64, 64)
    for i in range(len(ds)):
        assert (img[i].double() - cpu[i]["img"]).abs().max() < 1e-6
        assert torch.equal(mask[i], cpu[i]["mask"]) and torch.equal(depth[i], cpu[i]["depth"])
        if i < 10:
            assert np.abs(img[i].numpy() - GOLD["i%02d_img" % i]).max() < 1e-6
            assert np.array_equal(mask[i].numpy(), GOLD["i%02d_mask" % i]) and np.array_equal(depth[i].numpy(), GOLD["i%02d_depth" % i])
    for k in SCALARS:
        assert torch.equal(batch[k], torch.stack([c[k] for c in cpu]))


@pytest.mark.gpu
def test_loader_feeds_the_trainer_batch_shape(tmp_path):
    from scp_amd.data import data_loader
    opts = _opts(tmp_path, num_workers=2, img_size=96)
    np.random.seed(3)
    loader, ds = data_loader(opts)
    batches = list(loader)
    assert len(batches) == opts.total_iters
    b = batches[0]
    n = opts.batch_size * opts.repeat
    assert b["img"].shape == (n, 3, 96, 96) and b["img"].is_cuda and b["mask"].shape == (n, 1, 96, 96)
    assert b["foc_crop"].shape == (n, 2) and b["idx"].shape == (n, 1)
    assert 0 <= float(b["img"].min()) and float(b["img"].max()) <= 1 and set(b["mask"].unique().tolist()) <= {0.0, 1.0}


@pytest.mark.gpu
@pytest.mark.child_process
def test_trainer_train_loop_on_disk_dataset(tmp_path):
    """end to end: synthetic Wild6D directory -> data_loader (PIL decode on workers, HIP crop+resize) -> Trainer.train
    (the loop of model/trainer.py:104-125) -> checkpoint; finite losses, parameters move, checkpoint reloads"""
    import scenes
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.trainer import Trainer
    dino.ALLOW_RANDOM_INIT = True
    o = _opts(tmp_path)
    opts = Options("laptop_wild6d", batch_size=2, repeat=3, train=True, total_iters=4, img_size=256, ngpu=1, num_workers=2,
                   dataset_path=o.dataset_path, train_list=o.train_list, checkpoint_dir=str(tmp_path / "log"), name="t",
                   save_freq=4, batch_log_interval=2, local_rank=-1)
    np.random.seed(5)
    torch.manual_seed(0)
    tr = Trainer(opts, prior=scenes.bottle_like(3), device="cuda")
    before = tr.model.mesh.mean_v.detach().clone()
    lines = []
    history = tr.train(log=lines.append)
    assert len(history) == 4 and all(np.isfinite(history)) and len(lines) == 2
    assert not torch.equal(before, tr.model.mesh.mean_v)
    ckpt = os.path.join(str(tmp_path), "log", "t", "pred_net_4.pth")
    assert os.path.exists(ckpt)
    tr2 = Trainer(Options("laptop_wild6d", batch_size=2, repeat=3, train=True, total_iters=4, model_path=ckpt),
                  prior=scenes.bottle_like(3), device="cuda")
    assert torch.equal(tr2.model.mesh.mean_v, tr.model.mesh.mean_v)


def test_test_dataset_order_box_and_ground_truth(tmp_path):
    """Wild6DTestDataset (dataset_wild6d_test.py): frames in order with stride dframe_eval, fixed 1.35x box, gt from the pkl"""
    from scp_amd.data import Wild6DTestDataset
    root, list_path = wild6d_synth.write_test_set(str(tmp_path))
    o = types.SimpleNamespace(test_list=list_path, test_dataset_path=root, batch_size=2, img_size=64, use_depth=True, eval=True,
                              dframe_eval=2, no_stretch=False, ngpu=1, local_rank=-1, num_workers=0, shuffle_test=False, test=True)
    ds = Wild6DTestDataset(o)
    assert ds.sample_list == [(0, 0), (0, 2), (1, 0), (1, 2)]
    state = np.random.get_state()[1].copy()
    e = ds[1]
    assert np.array_equal(state, np.random.get_state()[1]), "the test item must not consume the numpy generator"
    assert e["img"].shape == (3, 64, 64) and e["rotation"].shape == (3, 3) and e["scale"].shape == (3,)
    assert int(e["idx"]) == 0 and int(e["frame_idx"]) == 2
    raw = ds.raw_item(1)
    ih, iw, pt, pl, vh, vw = raw["_crop"]["geom"]
    assert (vh, vw) == (2 * int(e["length"][1]), 2 * int(e["length"][0])) and torch.equal(raw["rotation"], e["rotation"])


@pytest.mark.gpu
@pytest.mark.child_process
def test_tester_test_loop_on_disk_test_set(tmp_path):
    """end to end: on-disk test set + pkl annotations -> test_loader -> Tester.test() -> pose metric table"""
    import scenes
    import scp_amd.dino as dino
    from scp_amd.flags import Options
    from scp_amd.tester import Tester
    dino.ALLOW_RANDOM_INIT = True
    root, list_path = wild6d_synth.write_test_set(str(tmp_path), n_frames=4, w=320, h=240)
    opts = Options("laptop_wild6d", batch_size=4, repeat=1, train=False, test=True, eval=True, eval_nocs=True, test_dataset_path=root,
                   test_list=list_path, num_workers=2, local_rank=-1, dframe_eval=1)
    t = Tester(opts, prior=scenes.bottle_like(3))
    torch.manual_seed(0)
    lines = []
    out = t.test(log=lines.append)
    assert out["n"] == 8 and len(t.deg_cm_result) == 8 and len(t.iou_result) == 8 and len(lines) == 6
    assert all(0.0 <= out[k] <= 1.0 for k in ("5deg2cm", "5deg5cm", "10deg2cm", "10deg5cm", "iou@25", "iou@50"))


@pytest.mark.gpu
def test_collator_staging_is_not_reused_while_a_copy_is_in_flight(tmp_path):
    """many batches collated back to back with a busy device: every batch must still equal a fresh collation of the same
    items (the pinned staging ring may only be refilled behind the event of the copy that read it)"""
    from scp_amd.data import GpuCollator
    ds = _dataset(tmp_path)
    np.random.seed(12)
    raws = [ds.raw_item(i) for i in range(len(ds))]
    groups = [raws[i:i + 6] for i in range(0, 24, 6)]
    ref = [GpuCollator(64, "cuda", True)(g)["img"].clone() for g in groups]
    torch.cuda.synchronize()
    col = GpuCollator(64, "cuda", True)
    busy = torch.randn(4096, 4096, device="cuda")
    outs = []
    for rep in range(6):
        for _ in range(3):
            busy = busy @ busy * 1e-3          # keep the stream behind the host
        for g in groups:
            outs.append(col(g)["img"])
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert torch.equal(o, ref[k % 4]), k
