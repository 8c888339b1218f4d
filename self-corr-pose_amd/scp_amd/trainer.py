"""scp_amd/trainer.py -- the training step.

The reference has no `step()`; its loop body is model/trainer.py:118-125 (zero_grad, forward,
`total_loss.mean().backward()`, collect_grad = per-group clipping + NaN guard, AdamW/OneCycle step).
`Trainer.step(data)` is exactly that body, minus tensorboard and the per-parameter host syncs
(SURVEY F15): the NaN guard is evaluated on device and applied by zeroing the gradients.
`batch_reshape` mirrors trainer.py:81-102 (NDC conversion of the crop intrinsics).
Data-parallel operation: scp_amd.parallel.GradientAllReducer averages gradients over RCCL before
the clip (the reference constructs DDP but bypasses it).
"""
import os
import tempfile

import torch

from . import streams
from .model import MeshNet
from .optimizers import Optimizers
from .parallel import FlatGradients


def freeze_batchnorm_affine(model):
    """trainer.py:54-58: BatchNorm2d weights/biases are frozen (statistics still update)"""
    for m in model.modules():
        if m.__class__.__name__ == "BatchNorm2d":
            for p in m.parameters():
                p.requires_grad = False


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "gemm_gfx950.csv")


def enable_gemm_tuning():
    """PyTorch's TunableOp for the few LIBRARY GEMMs the step still issues (the pose / shape / texture heads and the mesh encoder's
    linears, ~0.15 ms per step; every large product -- ViT, convolutions, correspondence -- runs on the build's own kernels).
    SCP_GEMM_TUNING = "read" (default): apply the selections recorded on an MI355X and shipped in tuning/gemm_gfx950.csv, never tune
    online (TunableOp ignores the file if its library-version validators do not match; the library heuristics then decide);
    "online": also time the candidates of shapes the file does not hold, during the first steps (what rounds 1-4 did by default: it
    makes the solution -- and the last bit of those products -- depend on the process, and runs every candidate kernel of the
    library once); "0": leave TunableOp alone.  New results go to a scratch file, never back into the package."""
    mode = os.environ.get("SCP_GEMM_TUNING", "read")
    if mode in ("0", "off"):
        return
    tun = torch.cuda.tunable
    tun.enable(True)
    # SCP_GEMM_TUNING_OUT=<file>: where this process writes what it ends up with (shipped + newly tuned shapes), for
    # regenerating tuning/gemm_gfx950.csv (tools/retune_gemms.sh); default is a scratch file
    out = os.environ.get("SCP_GEMM_TUNING_OUT") or os.path.join(tempfile.gettempdir(), "scp_tunableop_%d.csv" % os.getpid())
    tun.set_filename(out, False)
    tun.set_max_tuning_duration(30)
    tun.tuning_enable(mode in ("online", "1"))
    if os.path.exists(TUNED_GEMMS):
        tun.read_file(TUNED_GEMMS)


class Trainer:
    def __init__(self, opts, prior=None, device=None, process_group=None, sync_bn=False, graphs=None):
        self.opts = opts
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        # MIOpen solver search, as the reference does (train.py:21 cudnn.benchmark = True); without
        # it MIOpen's immediate mode falls back to naive fp32 convolutions for several layers
        torch.backends.cudnn.benchmark = True
        if self.device.type == "cuda":
            # the C-ABI launches go to the CURRENT device's current stream (scp_amd/capi.py): make the trainer's
            # device the current one so that Trainer(device="cuda:1") in a process sitting on cuda:0 cannot mix devices
            if self.device.index is not None:
                torch.cuda.set_device(self.device)
            enable_gemm_tuning()
        # the rotation-cycle branch runs on a side stream (model.py); its parameters' AccumulateGrad nodes
        # then see gradients from two streams, which autograd synchronises correctly but warns about
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self.model = MeshNet(opts, prior)
        if opts.model_path:
            self.model.load_network(opts.model_path)
        freeze_batchnorm_affine(self.model)
        # BatchNorm under data parallelism.  DEFAULT (sync_bn=False): every rank normalises with the statistics of its own 32
        # images -- the semantics the fused convolution + BatchNorm ops implement (scp_amd/fused_conv.py: the statistics are
        # finalised by the last workgroup of the rank's own convolution launch, no collective) and what the reference does on one
        # GPU; checkpoints carry the rank-averaged running statistics (save()).  sync_bn=True reproduces the reference's
        # multi-GPU choice (trainer.py:67 SyncBatchNorm.convert_sync_batchnorm): the modules become torch's SyncBatchNorm, which
        # the fused ops do not take (type check) -- those layers then run own convolution -> torch SyncBatchNorm (one stat
        # all-gather per layer over RCCL) -> ReLU as separate ops.
        self.sync_bn = bool(sync_bn)
        if sync_bn:
            self.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.model, process_group)
        self.model = self.model.to(self.device)
        if self.device.type == "cuda":
            self.model.encoder.backbone.to(memory_format=torch.channels_last)
            self.model.encoder.featnet.to(memory_format=torch.channels_last)
        self.model.train()
        self.optim = Optimizers(opts, self.model)
        self.iteration = 0
        self._steps_done = 0                            # steps run by THIS process (the first one is the solver-search step)
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        self._mean_v = [p for n, p in named if "mean_v" in n]
        self._shapenerf = [p for n, p in named if "mean_v" not in n and "shapenerf" in n]
        self._pose = [p for n, p in named if "mean_v" not in n and "shapenerf" not in n and "pose_predictor" in n]
        self._trainable = [p for _, p in named]
        # every gradient lives in one flat buffer (p.grad = view); with torch.distributed initialised its buckets are
        # all-reduced from autograd hooks while backward is still running (scp_amd/parallel.py)
        # SCP_FORCE_COLLECTIVES=1: the N > 1 code path (gradient hooks, communication stream, bucketed all-reduce inside backward) in a
        # ONE-rank group -- how the multi-GPU schedule of the step is exercised on a one-GPU box (tools/r06/hang_repro.py, tests)
        self.grads = FlatGradients(self._trainable, process_group, force_collectives=os.environ.get("SCP_FORCE_COLLECTIVES") == "1")
        if hasattr(self.optim.optimizer, "attach"):
            self.optim.optimizer.attach(self.grads)     # FlatAdamW: one launch over the flat buffer (scp_amd/optimizers.py)
        self.reducer = self.grads if self.grads.world > 1 else None
        self.rank = torch.distributed.get_rank(process_group) if torch.distributed.is_initialized() else 0
        if self.reducer is not None:
            # DDP's init broadcast (trainer.py:70-75): replicas start identical whatever the per-rank seeding, BatchNorm
            # buffers included
            self.reducer.broadcast_parameters(self.model, 0)
        self._group_spans = None
        # HIP-graph replay of the frozen ViT + pair matching (scp_amd/graphed.py GraphedInference; OPT-IN, SCP_GRAPHS=1 or graphs=True):
        # saves host enqueue time, not device time.  Not in configs[4] precision (autocast's weight-cast cache does not survive capture).
        from . import fused_conv
        self._convs = [m for m in self.model.encoder.modules() if isinstance(m, torch.nn.Conv2d)]
        fused_conv.WEIGHT_EPOCH[0] += 1                 # load_network / broadcast wrote weights through .data
        self.use_graphs = (self.device.type == "cuda" and not self.sync_bn and (graphs if graphs is not None else os.environ.get("SCP_GRAPHS", "0") == "1")
                           and not bool(getattr(opts, "mixed_bf16", False)))
        self.model.pretrain_corr_net.use_graphs = self.use_graphs

    def batch_reshape(self, batch):
        o, dev = self.opts, self.device
        img = batch["img"].float().to(dev, non_blocking=True)
        mask = batch["mask"].to(dev, non_blocking=True).squeeze(1)
        depth = batch["depth"].to(dev, non_blocking=True).squeeze(1) if o.use_depth else None
        occ = batch["occ"].to(dev, non_blocking=True).squeeze(1) if o.use_occ else None
        to = lambda k: batch[k].to(dev, non_blocking=True)
        pp_crop = to("pp_crop") / (o.img_size / 2.) - 1.
        foc_crop = to("foc_crop") / (o.img_size / 2.)
        return (img, mask, depth, occ, batch["center"], batch["length"], to("foc"), foc_crop, to("pp"), pp_crop,
                to("idx"), None)

    def collect_grad(self):
        """All-reduce (data parallel), per-group clipping (trainer.py:132-150: mean_v 1.0, shapenerf 1.0,
        pose_predictor 0.1) and NaN guard (a non-finite gradient anywhere zeroes every gradient, like
        the reference's zero_grad()) directly on the flat gradient buffer the parameters' .grad are views of:
        ~10 launches, no host round trip, no gather / scatter copies (the reference does one isnan().sum() > 0
        host sync per parameter).  With N > 1 most of the all-reduce has already run underneath backward."""
        # SUM over ranks; waits for the in-flight buckets.  Parameters without a gradient this step (shapenerf under no_deform,
        # decoder units a flag switches off) keep grad = None like after the reference's zero_grad(): AdamW skips them
        flat = self.grads.finish(keep_unused_none=True)
        if self._group_spans is None:
            self._group_spans = []
            for group, max_norm in ((self._mean_v, 1.), (self._shapenerf, 1.), (self._pose, 0.1)):
                spans = sorted(self.grads.span[id(p)] for p in group)
                merged = []
                for o, n in spans:                       # merge adjacent parameters into contiguous ranges
                    if merged and merged[-1][0] + merged[-1][1] == o:
                        merged[-1] = (merged[-1][0], merged[-1][1] + n)
                    else:
                        merged.append((o, n))
                self._group_spans.append((merged, max_norm))
        fused = self._fused_clip(flat)
        if fused is not None:
            return fused
        if self.grads.world > 1:
            flat.div_(self.grads.world)
        finite = torch.isfinite(flat).all()
        flat.nan_to_num_(0., 0., 0.).mul_(finite.to(flat.dtype))
        out = []
        for spans, max_norm in self._group_spans:
            if not spans:
                out.append(torch.zeros((), device=self.device))
                continue
            seg = [flat[o:o + n] for o, n in spans]
            total = seg[0].norm(2) if len(seg) == 1 else torch.stack([t.norm(2) for t in seg]).norm(2)
            coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)           # clip_grad_norm_'s coefficient
            for t in seg:
                t.mul_(coef)
            out.append(total)
        return tuple(out)

    def _fused_clip(self, flat):
        """the same on the GPU as two launches over the flat buffer (csrc/gradclip.hip: reduction with the NaN flag, then
        g <- finite ? coef[group] * g / world : 0) instead of ~25 torch launches and eight passes; None = not applicable"""
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and getattr(self, "fuse_clip", True)):
            return None
        import ctypes
        from . import capi
        ranges = [(o, o + n, gi) for gi, (spans, _) in enumerate(self._group_spans) for o, n in spans]
        L = capi.lib()
        if len(ranges) > 16:
            return None
        if getattr(self, "_clip_ws", None) is None:
            nb = L.scp_gradclip_workspace()
            self._clip_ws = torch.zeros(nb, dtype=torch.uint8, device=flat.device)
            arr = ctypes.c_longlong * max(len(ranges), 1)
            self._clip_args = (arr(*[r[0] for r in ranges]), arr(*[r[1] for r in ranges]), (ctypes.c_int * max(len(ranges), 1))(*[r[2] for r in ranges]),
                               len(ranges), self._clip_ws.numel())
        begin, end, group, n, nb = self._clip_args
        result = torch.empty(7, dtype=torch.float32, device=flat.device)
        mx = [m for _, m in self._group_spans]
        capi.check(L.scp_gradclip(capi.dev_ptr(flat, "flat"), flat.numel(), 1.0 / self.grads.world, begin, end, group, n, mx[0], mx[1], mx[2],
                                  ctypes.c_void_p(self._clip_ws.data_ptr()), nb, capi.dev_ptr(result, "result"), capi.current_stream()),
                   "scp_gradclip")
        self.last_clip = result
        return result[0], result[1], result[2]

    def step(self, data, next_data=None):
        """one training iteration on an already device-resident 12-tuple; returns (total_loss, aux, grad norms).
        `next_data` (optional): the following batch, if the input pipeline already has it on the device.
        Its frozen-DINO features depend on nothing but the images, so their computation is enqueued on the
        side stream before this step's backward and overlaps with it (software pipelining across
        iterations; the next step finds them ready).  Per-step work is unchanged."""
        self.model.iters = self.iteration
        self.grads.prepare()                            # zero_grad: clears the flat buffer, p.grad = its views
        if self.device.type == "cuda":
            # the split operands of the convolution weights the optimizer just changed: rebuilt here, on the main stream, before any
            # side stream or graph replay reads them (scp_amd/fused_conv.py weight_planes)
            from . import fused_conv
            fused_conv.refresh_planes(self._convs)
        # The first step of a process is where MIOpen's solver search (cudnn.benchmark) times its candidates for every
        # convolution shape, forward and backward.  It runs without the side streams, so that the search measures
        # undisturbed kernels instead of kernels sharing the device with the ViT / the second encoder pass (the winners
        # are cached per process; a perturbed search can settle on slower solvers for the whole run).
        serial = (self._steps_done == 0 or not streams.overlap()) and self.device.type == "cuda"
        if serial:
            saved = tuple(getattr(self.model, k, d) for k, d in (("overlap_dino", streams.overlap()), ("overlap_rotation_cycle", streams.overlap()),
                                                                 ("overlap_texture_pass", streams.overlap_texture())))
            self.model.overlap_dino = self.model.overlap_rotation_cycle = self.model.overlap_texture_pass = False
            next_data = None
        try:
            streams.crumb("step %d: start" % self.iteration)
            total_loss, aux_output = self.model(data)
            streams.crumb("step %d: forward done" % self.iteration)
            if next_data is not None:
                self.model.pretrain_corr_net.prefetch_features(next_data[0], next_data[1])
            total_loss.mean().backward()
            streams.crumb("step %d: backward done" % self.iteration)
        finally:
            if serial:
                self.model.overlap_dino, self.model.overlap_rotation_cycle, self.model.overlap_texture_pass = saved
        self._steps_done += 1
        grad = self.collect_grad()
        streams.crumb("step %d: clip done" % self.iteration)
        self.optim.step(self.iteration)
        streams.crumb("step %d: optimizer done" % self.iteration)
        self.iteration += 1
        return total_loss, aux_output, grad

    def train(self, loader=None, log=print):
        """the loop of model/trainer.py:104-125: data_loader -> batch_reshape -> step -> periodic log / checkpoint.
        `loader` defaults to scp_amd.data.data_loader(opts) (same Wild6D layout and sampler, resize on the device);
        any iterable of reference-style batch dicts works.  Logging is a one-line print (the reference's tensorboard
        writer and visualisation block are out of scope); checkpoints go to <checkpoint_dir>/<name>/pred_net_<it>.pth
        every `save_freq` iterations like trainer.py:152-158.  Returns the list of per-iteration total losses."""
        import time
        opts = self.opts
        if loader is None:
            from .data import data_loader
            loader, self.dataset = data_loader(opts, self.device)
        save_dir = os.path.join(opts.checkpoint_dir, opts.name)
        history, t0 = [], time.time()
        pending = []                                   # device scalars; read back only at log time (no per-step sync)
        # one batch of look-ahead: the frozen-DINO features of batch i+1 are enqueued during the backward of batch i (step())
        it = iter(loader)
        nxt = next(it, None)
        nxt_dev = self.batch_reshape(nxt) if nxt is not None else None
        i = -1
        while nxt is not None:
            i += 1
            batch, data = nxt, nxt_dev
            nxt = next(it, None)
            nxt_dev = self.batch_reshape(nxt) if nxt is not None else None
            total, aux, grad = self.step(data, next_data=nxt_dev)
            pending.append(total.detach())
            if (i + 1) % opts.batch_log_interval == 0:
                vals = self._read_back(pending, "Trainer.train: losses of iterations %d..%d" % (i + 2 - len(pending), i + 1))
                self.grads.check_static_graph()        # the host is synchronised here anyway
                history.extend(vals)
                pending = []
                t1 = time.time()
                log("batch %d, batch size %d, mean per iter time:%.4f, loss %.5f" % (
                    i + 1, batch["img"].shape[0], (t1 - t0) / opts.batch_log_interval, vals[-1]))
                t0 = t1
            if opts.save_freq and (i + 1) % opts.save_freq == 0:
                self.save(os.path.join(save_dir, "pred_net_%d.pth" % (i + 1)))
        if pending:
            history.extend(self._read_back(pending, "Trainer.train: losses of the last %d iterations" % len(pending)))
        return history

    def named_streams(self):
        """the HIP streams a step enqueues on, by role (for DeviceStall reports)"""
        if self.device.type != "cuda":
            return {}
        m = self.model
        out = {"main": torch.cuda.current_stream(self.device),
               "rotation-cycle side stream": getattr(m, "_cycle_stream", None), "texture-pass side stream": getattr(m, "_tex_stream", None),
               "frozen-ViT side stream": getattr(m.pretrain_corr_net, "_side_stream", None),
               "all-reduce stream": getattr(self.grads, "comm_stream", None)}
        return {k: v for k, v in out.items() if v is not None}

    def _read_back(self, pending, what):
        """device scalars -> host floats; the wait for the device is bounded (streams.wait_bounded raises DeviceStall)"""
        stacked = torch.stack(pending)
        if stacked.is_cuda:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            streams.wait_bounded(done, what, self.named_streams)
        return stacked.cpu().tolist()

    def save(self, path):
        """rank 0 writes (trainer.py:152-158 guards with local_rank <= 0).  COLLECTIVE when world > 1: every rank must call it
        (Trainer.train does) -- the per-rank BatchNorm running statistics are averaged INTO THE SAVED COPY so the checkpoint
        does not depend on which rank writes it; the live buffers keep their per-rank values and nothing else is reduced."""
        averaged = self.reducer.averaged_running_stats(self.model) if self.reducer is not None else {}
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        state = dict(self.model.state_dict())
        for name, avg in averaged.items():
            if name in state:
                state[name] = avg
        state["mesh.faces"] = self.model.mesh.faces.cpu()
        torch.save(state, path)
