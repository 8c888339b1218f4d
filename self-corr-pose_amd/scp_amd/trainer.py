"""scp_amd/trainer.py -- the training step.

The reference has no `step()`; its loop body is model/trainer.py:118-125 (zero_grad, forward,
`total_loss.mean().backward()`, collect_grad = per-group clipping + NaN guard, AdamW/OneCycle step).
`Trainer.step(data)` is exactly that body, minus tensorboard and the per-parameter host syncs
(SURVEY F15): the NaN guard is evaluated on device and applied by zeroing the gradients.
`batch_reshape` mirrors trainer.py:81-102 (NDC conversion of the crop intrinsics).
Data-parallel operation: scp_amd.parallel.GradientAllReducer averages gradients over RCCL before
the clip (the reference constructs DDP but bypasses it).
"""
import torch

from .model import MeshNet
from .optimizers import Optimizers
from .parallel import GradientAllReducer


def freeze_batchnorm_affine(model):
    """trainer.py:54-58: BatchNorm2d weights/biases are frozen (statistics still update)"""
    for m in model.modules():
        if m.__class__.__name__ == "BatchNorm2d":
            for p in m.parameters():
                p.requires_grad = False


class Trainer:
    def __init__(self, opts, prior=None, device=None, process_group=None, sync_bn=False):
        self.opts = opts
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        # MIOpen solver search, as the reference does (train.py:21 cudnn.benchmark = True); without
        # it MIOpen's immediate mode falls back to naive fp32 convolutions for several layers
        torch.backends.cudnn.benchmark = True
        self.model = MeshNet(opts, prior)
        if opts.model_path:
            self.model.load_network(opts.model_path)
        freeze_batchnorm_affine(self.model)
        if sync_bn:
            self.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.model, process_group)
        self.model = self.model.to(self.device)
        if self.device.type == "cuda":
            self.model.encoder.backbone.to(memory_format=torch.channels_last)
            self.model.encoder.featnet.to(memory_format=torch.channels_last)
        self.model.train()
        self.optim = Optimizers(opts, self.model)
        self.reducer = GradientAllReducer(self.model, process_group) if torch.distributed.is_initialized() else None
        self.iteration = 0
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        self._mean_v = [p for n, p in named if "mean_v" in n]
        self._shapenerf = [p for n, p in named if "mean_v" not in n and "shapenerf" in n]
        self._pose = [p for n, p in named if "mean_v" not in n and "shapenerf" not in n and "pose_predictor" in n]
        self._trainable = [p for _, p in named]

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
        """per-group clipping of trainer.py:132-150; a non-finite gradient anywhere drops the step
        (all gradients zeroed) without a host round trip"""
        grads = [p.grad for p in self._trainable if p.grad is not None]
        finite = torch.stack([g.isfinite().all() for g in grads]).all() if grads else None
        norms = []
        for group, max_norm in ((self._mean_v, 1.), (self._shapenerf, 1.), (self._pose, 0.1)):
            ps = [p for p in group if p.grad is not None]
            norms.append(torch.nn.utils.clip_grad_norm_(ps, max_norm) if ps else torch.zeros((), device=self.device))
        if finite is not None:
            keep = finite.to(grads[0].dtype)
            torch._foreach_mul_(grads, keep)
            for g in grads:   # 0 * nan = nan: scrub
                torch.nan_to_num_(g, nan=0.0, posinf=0.0, neginf=0.0)
        return tuple(norms)

    def step(self, data):
        """one training iteration on an already device-resident 12-tuple; returns (total_loss, aux, grad norms)"""
        self.model.iters = self.iteration
        self.optim.zero_grad()
        total_loss, aux_output = self.model(data)
        total_loss.mean().backward()
        if self.reducer is not None:
            self.reducer.all_reduce()
        grad = self.collect_grad()
        self.optim.step(self.iteration)
        self.iteration += 1
        return total_loss, aux_output, grad

    def save(self, path):
        state = self.model.state_dict()
        state["mesh.faces"] = self.model.mesh.faces.cpu()
        torch.save(state, path)
