"""scp_amd/optimizers.py -- AdamW + OneCycleLR with the reference's name-based parameter groups
(model/module/optimizers.py:5-84): mean_v / pose_predictor / shape(_code)_predictor / featnet /
backbone get vert_lr_ratio, cam_lr_ratio, 1, 1, 1 x learning_rate; pretrain_corr_net is frozen.
`fused=True` on GPU keeps the update in one multi-tensor launch."""
import torch

GROUPS = ("mean_v", "pose_predictor", "shape", "featnet", "backbone")


def group_of(name):
    if "mean_v" in name:
        return 0
    if "pose_predictor" in name:
        return 1
    if "shape_predictor" in name or "shape_code_predictor" in name:
        return 2
    if "featnet" in name:
        return 3
    if "backbone" in name:
        return 4
    return None


class Optimizers:
    def __init__(self, opts, model):
        self.opts, self.model = opts, model
        # the reference sizes the schedule with total_iters * ngpu although every rank steps
        # total_iters times; kept for schedule parity
        self.total_steps = opts.total_iters * opts.ngpu
        groups = [[] for _ in GROUPS]
        for name, p in model.named_parameters():
            g = group_of(name)
            if g is not None:
                groups[g].append(p)
        lr = opts.learning_rate
        on_gpu = any(p.is_cuda for g in groups for p in g)
        self.optimizer = torch.optim.AdamW([{"params": g} for g in groups], lr=lr, betas=(0.9, 0.999),
                                           weight_decay=1e-4, **({"fused": True} if on_gpu else {}))
        if on_gpu:
            # an optimizer may step through .data (no version bump on the parameter): the convolutions' cached split planes are keyed
            # by tensor version AND this epoch (scp_amd/fused_conv.py), so the step itself invalidates them whatever its implementation
            from . import fused_conv
            self.optimizer.register_step_post_hook(lambda *a, **k: fused_conv.invalidate())
        max_lrs = [opts.vert_lr_ratio * lr, opts.cam_lr_ratio * lr, lr, lr, lr]
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
            self.optimizer, max_lrs, total_steps=self.total_steps, pct_start=0.05, cycle_momentum=False,
            anneal_strategy="cos", final_div_factor=25, div_factor=25)

    def step(self, it=None):
        self.optimizer.step()
        self.scheduler.step()

    def zero_grad(self):
        self.optimizer.zero_grad(set_to_none=True)
