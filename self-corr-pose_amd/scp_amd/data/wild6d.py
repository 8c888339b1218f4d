"""scp_amd/data/wild6d.py -- Wild6D training input pipeline (data/dataset_wild6d.py:37-182, data/dataloader.py:52-66).

Same on-disk format, same sampler (`reset`: np.random video / frame draws per iteration, :99-112), same per-item
quantities (mask box, 1.2-1.5x random box scale, crop factors, NDC-ready crop intrinsics, :115-182), same batch keys.
What changes is where the pixels are resized:

  reference   every worker: cv2 decode -> float64 image -> torchvision resized_crop to 256x256 (float64 bilinear) ->
              collate -> pageable H2D of [B,3,256,256] float64 + mask + depth
  here        every worker: PIL decode -> cut the uint8 crop box (`raw_item`) -> the batch's crops laid out in ONE uint8 staging
              tensor (`stage_batch`: shared memory across the process boundary, pinned by the loader's pin thread);  main
              process: one async H2D, ONE HIP launch for the whole batch (csrc/crop_resize.hip) -> float32 CUDA tensors
              (`GpuCollator`); the trainer's `batch_reshape` consumes them unchanged.

`Wild6DDataset.__getitem__` keeps the reference's CPU semantics (float64 image through F.interpolate, which is what
torchvision's tensor backend calls) for drop-in use and as the comparison path.  cv2 and torchvision are un-vendored
dependencies that are absent here: decode is PIL's (JPEG IDCTs may differ from libjpeg-turbo's in the last bit),
crop/resize follow torchvision 0.11's published tensor implementation -- parity unpinned at those two calls, pinned
for everything around them by tests/golden/wild6d_items.npz (recorded from the reference class itself)."""
import ctypes
import glob
import json
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from .. import capi


def _imread_rgb(path):
    return np.array(Image.open(path).convert("RGB"))


def _imread_gray(path):
    return np.array(Image.open(path).convert("L"))


def _imread_unchanged(path):
    return np.array(Image.open(path))


def _resized_crop(img, top, left, height, width, size, mode):
    """torchvision 0.11 functional_tensor: crop (zero padding outside the frame), then interpolate"""
    h, w = img.shape[-2:]
    right, bottom = left + width, top + height
    if left < 0 or top < 0 or right > w or bottom > h:
        pad = [max(-left, 0), max(right - w, 0), max(-top, 0), max(bottom - h, 0)]
        img = F.pad(img[..., max(top, 0):bottom, max(left, 0):right], pad)
    else:
        img = img[..., top:bottom, left:right]
    kw = dict(align_corners=False) if mode == "bilinear" else {}
    return F.interpolate(img[None], size=list(size), mode=mode, **kw)[0]


class Wild6DDataset(Dataset):
    training = True

    def __init__(self, opts):
        self.opts = opts
        self._index(opts.train_list, opts.dataset_path)
        self.samples_per_iter = opts.batch_size * opts.repeat * opts.ngpu
        self.samples_total = opts.total_iters * self.samples_per_iter
        self.sample_list = None
        self.reset()

    def _index(self, list_path, root):
        """sequence list -> per-sequence frame lists and intrinsics (:44-75; the test set is laid out the same way)"""
        with open(list_path) as f:
            self.train_list = f.read().strip().split()
        self.imglist, self.masklist, self.depthlist, self.metalist, self.seq_names = [], [], [], [], []
        self.total_frames = 0
        obj_list = sorted(os.listdir(root))
        for seqname in self.train_list:
            parts = seqname.split("_")
            obj_index, seq_index = int(parts[-2]), int(parts[-1])
            seq_list = sorted(os.listdir(os.path.join(root, obj_list[obj_index])))
            seq_dir = os.path.join(root, obj_list[obj_index], seq_list[seq_index])
            self.seq_names.append((obj_list[obj_index], seq_list[seq_index]))
            mask_list = glob.glob(os.path.join(seq_dir, "images/*-mask.png"))
            mask_list.sort(key=lambda item: int(item.split("/")[-1].split("-")[0]))
            self.masklist.append(mask_list)
            self.imglist.append([m.replace("-mask.png", ".jpg") for m in mask_list])
            self.depthlist.append([m.replace("-mask.png", "-depth.png") for m in mask_list])
            with open(os.path.join(seq_dir, "metadata"), "rb") as f:
                meta = json.load(f)
            K = np.array(meta["K"]).reshape(3, 3).T if "K" in meta else None     # stored column-major
            self.metalist.append((K, meta.get("w"), meta.get("h"), meta.get("fps")))
            self.total_frames += len(mask_list)

    def __len__(self):
        return self.samples_total

    def reset(self):
        """the sampler: per iteration `batch_size` videos, per video `repeat` strata x `ngpu` frames (:99-112)"""
        o = self.opts
        total = []
        for _ in range(o.total_iters):
            videos = np.random.randint(0, len(self.masklist), size=(o.batch_size,))
            frames = []
            for v in videos:
                n_gap = len(self.masklist[v]) // o.repeat
                for i in range(o.repeat):
                    for _ in range(o.ngpu):
                        frames.append((v, n_gap * i + np.random.randint(0, n_gap)))
            total.append(frames)
        self.sample_list = total

    # ---- the part shared by both paths: decode, box, intrinsics (:115-158) -------------------------------------------
    def _sample(self, index):
        """(video, frame, box scale) of item `index`: training draws the 1.2-1.5x scale per item (:121)"""
        batch_id, item_id = divmod(index, self.samples_per_iter)
        video_id, frame_id = self.sample_list[batch_id][item_id]
        return video_id, frame_id, np.random.uniform(1.2, 1.5, size=(2,))

    def _load(self, index):
        o = self.opts
        video_id, frame_id, rand_scale = self._sample(index)
        img = _imread_rgb(self.imglist[video_id][frame_id])
        mask = _imread_gray(self.masklist[video_id][frame_id]).astype(bool)
        depth = _imread_unchanged(self.depthlist[video_id][frame_id]) if o.use_depth else None
        intr = self.metalist[video_id][0]
        yid, xid = np.where(mask > 0)
        center = [(xid.max() + xid.min()) // 2, (yid.max() + yid.min()) // 2]
        length = [(xid.max() - xid.min()) // 2, (yid.max() - yid.min()) // 2]
        if o.no_stretch and self.training:
            m = max(length)
            length = [int(rand_scale[0] * m), int(rand_scale[0] * m)]
        else:
            length = [int(rand_scale[0] * length[0]), int(rand_scale[1] * length[1])]
        foc, pp = [intr[0, 0], intr[1, 1]], [intr[0, 2], intr[1, 2]]
        s = o.img_size
        crop_factor = [s / 2 / length[0], s / 2 / length[1]]
        foc_crop = [foc[0] * crop_factor[0], foc[1] * crop_factor[1]]
        pp_crop = [(pp[0] - (center[0] - length[0])) * crop_factor[0], (pp[1] - (center[1] - length[1])) * crop_factor[1]]
        scalars = {
            "center": torch.tensor(center), "length": torch.tensor(length), "foc": torch.tensor(foc),
            "foc_crop": torch.tensor(foc_crop), "pp": torch.tensor(pp), "pp_crop": torch.tensor(pp_crop),
            "idx": torch.tensor([video_id]), "frame_idx": torch.tensor([frame_id]),
        }
        box = (int(center[1] - length[1]), int(center[0] - length[0]), int(2 * length[1]), int(2 * length[0]))  # top, left, h, w
        return img, mask, depth, box, scalars

    def __getitem__(self, index):
        """the reference's item: float64 [3,S,S] image in [0,1], float32 mask / depth [1,S,S]"""
        o = self.opts
        img, mask, depth, (top, left, bh, bw), elem = self._load(index)
        size = (o.img_size, o.img_size)
        img_t = torch.from_numpy(np.ascontiguousarray((img * 1.0).transpose(2, 0, 1))) / 255.
        elem["img"] = _resized_crop(img_t, top, left, bh, bw, size, "bilinear")
        elem["mask"] = _resized_crop(torch.tensor(mask, dtype=torch.float32)[None], top, left, bh, bw, size, "nearest")
        if o.use_depth:
            elem["depth"] = _resized_crop(torch.tensor(depth * 1.0, dtype=torch.float32)[None], top, left, bh, bw, size, "nearest")
        else:
            elem["depth"] = torch.zeros(1)
        return elem

    def raw_item(self, index):
        """the worker-side half of the device path: the in-frame part of the crop box as uint8 / uint16, the box
        geometry, and the same scalar fields"""
        img, mask, depth, (top, left, bh, bw), elem = self._load(index)
        h, w = mask.shape
        y0, y1, x0, x1 = max(top, 0), min(top + bh, h), max(left, 0), min(left + bw, w)
        elem["_crop"] = {
            "img": np.ascontiguousarray(img[y0:y1, x0:x1]),
            "mask": np.ascontiguousarray(mask[y0:y1, x0:x1].astype(np.uint8)),
            "depth": None if depth is None else np.ascontiguousarray(depth[y0:y1, x0:x1].astype(np.uint16)),
            "geom": (y1 - y0, x1 - x0, y0 - top, x0 - left, bh, bw),   # in_h, in_w, pad_top, pad_left, virt_h, virt_w
        }
        return elem


class Wild6DTestDataset(Wild6DDataset):
    """data/dataset_wild6d_test.py:37-200: every `dframe_eval`-th frame of every listed sequence in order, a fixed 1.35x box,
    and -- with opts.eval -- the ground-truth pose of each frame from
    <test_set>/../pkl_annotations/<class>/<class>-<obj>-<seq>.pkl (`annotations[i]` = {name, rotation, translation, size})"""
    training = False

    def __init__(self, opts):
        import pickle
        self.opts = opts
        self._index(opts.test_list, opts.test_dataset_path)
        self.rot_gt_list, self.trans_gt_list, self.scale_gt_list = [], [], []
        for obj, seq in self.seq_names:
            rots, transs, sizes = [], [], []
            if getattr(opts, "eval", False):
                root = opts.test_dataset_path
                prefix = root.rfind("test_set") + 9
                class_name = root[prefix:-1]
                gt_path = root[:prefix] + "pkl_annotations/" + root[prefix:] + "{}-{}-{}.pkl".format(class_name, obj, seq)
                with open(gt_path, "rb") as f:
                    gt = pickle.load(f)
                for i, anno in enumerate(gt["annotations"]):
                    assert int(anno["name"].split("/")[3]) == i
                    rots.append(np.array(anno["rotation"]))
                    transs.append(np.array(anno["translation"]))
                    sizes.append(np.array(anno["size"]))
            self.rot_gt_list.append(rots)
            self.trans_gt_list.append(transs)
            self.scale_gt_list.append(sizes)
        step = max(1, int(getattr(opts, "dframe_eval", 1)))
        self.sample_list = [(v, i) for v in range(len(self.masklist)) for i in range(0, len(self.masklist[v]), step)]

    def __len__(self):
        return len(self.sample_list)

    def reset(self):
        pass

    def _sample(self, index):
        video_id, frame_id = self.sample_list[index]
        return video_id, frame_id, np.array([1.35, 1.35])

    def _load(self, index):
        out = super()._load(index)
        if getattr(self.opts, "eval", False):
            video_id, frame_id = self.sample_list[index]
            elem = out[4]
            elem["rotation"] = torch.tensor(self.rot_gt_list[video_id][frame_id])
            elem["translation"] = torch.tensor(self.trans_gt_list[video_id][frame_id])
            elem["scale"] = torch.tensor(self.scale_gt_list[video_id][frame_id])
        return out


class _RawView(Dataset):
    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        return self.ds.raw_item(i)


class GpuCollator:
    """list of raw items -> the reference's batch dict with `img` / `mask` / `depth` already on the device.  One pinned
    staging buffer (grown on demand, reused), one descriptor upload, one launch; everything is queued on the current
    stream.  No CPU fallback: without libscp_hip.so this raises."""

    RING = 3   # staging buffers in flight: the host fills buffer k+1 while the copy engine still reads buffer k

    def __init__(self, img_size, device="cuda", use_depth=True):
        self.size, self.device, self.use_depth = img_size, torch.device(device), use_depth
        self._ring = [[None, None] for _ in range(self.RING)]      # [pinned buffer, event recorded after its H2D copy]
        self._next = 0

    def _staging(self, nbytes):
        """a pinned staging buffer that no asynchronous copy is still reading: the copy is enqueued with non_blocking=True,
        so the buffer may only be refilled once the event recorded behind that copy has completed"""
        slot = self._ring[self._next]
        self._next = (self._next + 1) % self.RING
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < nbytes:
            slot[0] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        return slot

    def __call__(self, items):
        """`items`: a list of raw items (staged here), or the dict a worker already staged (stage_batch)"""
        L = capi.lib()
        st = items if isinstance(items, dict) else stage_batch(items, self.use_depth)
        B, S, off, staged = st["_B"], self.size, st["_off"], st["_staged"]
        nbytes = staged.numel()
        if staged.is_pinned():
            # the DataLoader's pin thread already moved the worker's buffer into page-locked memory: upload it as it is (the
            # caching host allocator keeps the block alive until the copy has completed)
            dev_buf = staged.to(self.device, non_blocking=True)
        else:
            slot = self._staging(nbytes)
            slot[0][:nbytes].copy_(staged)
            dev_buf = slot[0][:nbytes].to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(self.device))
        img = torch.empty(B, 3, S, S, dtype=torch.float32, device=self.device)
        mask = torch.empty(B, 1, S, S, dtype=torch.float32, device=self.device)
        depth = torch.empty(B, 1, S, S, dtype=torch.float32, device=self.device) if self.use_depth else None
        base = dev_buf.data_ptr()
        capi.check(L.scp_crop_resize_batch(ctypes.c_void_p(base), ctypes.c_void_p(base + off), B, S, ctypes.c_void_p(img.data_ptr()),
                                           ctypes.c_void_p(mask.data_ptr()),
                                           ctypes.c_void_p(depth.data_ptr() if depth is not None else 0),
                                           capi.current_stream()), "crop_resize_batch")
        dev_buf.record_stream(torch.cuda.current_stream(self.device))
        batch = {k: v for k, v in st.items() if not k.startswith("_")}
        batch["img"], batch["mask"] = img, mask
        batch["depth"] = depth if self.use_depth else torch.zeros(B, 1)
        return batch


def stage_batch(items, use_depth=True):
    """raw items -> ONE contiguous uint8 tensor in the staging layout scp_crop_resize_batch reads (per item: image, mask, [depth],
    16-byte aligned; then the descriptor table) + the stacked scalar fields.  Runs in the DataLoader WORKERS (`StageCollate`): what
    crosses the process boundary is then one shared-memory tensor per batch instead of ~100 pickled numpy arrays (the pipe copy of
    0.6 MB per image capped a rank at ~1000 images/s), and the main process only uploads it."""
    B = len(items)
    descs = (capi.CropDesc * B)()
    off = 0
    plan = []
    for i, it in enumerate(items):
        c = it["_crop"]
        ih, iw, pt, pl, vh, vw = c["geom"]
        d = descs[i]
        d.in_h, d.in_w, d.pad_top, d.pad_left, d.virt_h, d.virt_w = ih, iw, pt, pl, vh, vw
        d.img_off = off; off += ih * iw * 3
        d.mask_off = off; off += ih * iw
        off += off & 1                                          # uint16 alignment
        d.depth_off = off
        if use_depth:
            off += ih * iw * 2
        off = (off + 15) & ~15
        plan.append((d.img_off, d.mask_off, d.depth_off, c))
    desc_bytes = ctypes.sizeof(capi.CropDesc) * B
    staged = torch.empty(off + desc_bytes, dtype=torch.uint8)
    flat = staged.numpy()
    for img_off, mask_off, depth_off, c in plan:
        flat[img_off:img_off + c["img"].size] = c["img"].reshape(-1)
        flat[mask_off:mask_off + c["mask"].size] = c["mask"].reshape(-1)
        if use_depth:
            flat[depth_off:depth_off + c["depth"].size * 2] = c["depth"].reshape(-1).view(np.uint8)
    flat[off:off + desc_bytes] = np.frombuffer(bytes(descs), dtype=np.uint8)
    out = {k: torch.stack([it[k] for it in items]) for k in items[0] if k != "_crop"}
    out.update(_staged=staged, _off=off, _B=B)
    return out


class StageCollate:
    """picklable collate_fn for the DataLoader workers"""

    def __init__(self, use_depth):
        self.use_depth = use_depth

    def __call__(self, items):
        return stage_batch(items, self.use_depth)


class _DeviceLoader:
    def __init__(self, loader, collator):
        self.loader, self.collator = loader, collator

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            yield self.collator(items)


def test_loader(opts, device="cuda"):
    """data/dataloader.py:69-84 (batch_size = opts.batch_size, no drop_last) with the resize on the device"""
    dataset = Wild6DTestDataset(opts)
    sampler = None
    if getattr(opts, "local_rank", -1) != -1 and getattr(opts, "ngpu", 1) > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=opts.ngpu, rank=opts.local_rank, shuffle=False)
    loader = DataLoader(_RawView(dataset), batch_size=opts.batch_size, num_workers=getattr(opts, "num_workers", 0), sampler=sampler,
                        shuffle=bool(getattr(opts, "shuffle_test", False)) and sampler is None, collate_fn=StageCollate(opts.use_depth),
                        pin_memory=str(device).startswith("cuda"))
    return _DeviceLoader(loader, GpuCollator(opts.img_size, device, opts.use_depth)), dataset


def data_loader(opts, device="cuda"):
    """data/dataloader.py:52-66 with the resize moved to the device.  Returns (iterable of batch dicts, dataset)."""
    dataset = Wild6DDataset(opts)
    sampler = None
    if getattr(opts, "local_rank", -1) != -1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=opts.ngpu, rank=opts.local_rank, shuffle=False)
    loader = DataLoader(_RawView(dataset), batch_size=opts.batch_size * opts.repeat, num_workers=getattr(opts, "num_workers", 0),
                        drop_last=True, sampler=sampler, shuffle=False, collate_fn=StageCollate(opts.use_depth),
                        pin_memory=str(device).startswith("cuda"))
    return _DeviceLoader(loader, GpuCollator(opts.img_size, device, opts.use_depth)), dataset
