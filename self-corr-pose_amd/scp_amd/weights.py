"""scp_amd/weights.py -- loss-weight schedule (model/module/weights.py:21-64): regularisers decay
linearly to decay_ratio x their value over total_iters, match/imatch ramp the other way."""
import numpy as np


def reg_decay(curr_steps, max_steps, min_wt, max_wt, mode="linear"):
    if curr_steps > max_steps:
        return min_wt
    frac = curr_steps / float(max_steps)
    if mode == "log":
        return np.exp(frac * (np.log(min_wt) - np.log(max_wt))) * max_wt
    if mode == "linear":
        return frac * (min_wt - max_wt) + max_wt
    raise NotImplementedError


class Weights:
    _FIXED = ("mask_wt", "depth_wt", "tex_wt", "match_wt", "imatch_wt", "triangle_wt", "pullfar_wt",
              "deform_wt", "symmetry_wt", "camera_wt", "cycle_loss_wt")

    def __init__(self, opts):
        self.opts = opts
        self.total_iters = opts.total_iters
        for name in self._FIXED:
            setattr(self, name, getattr(opts, name))
        self.cycle_loss_pt_wt = opts.cycle_loss_pretrain_wt

    def schedule(self, it):
        o, r, n = self.opts, self.opts.decay_ratio, self.total_iters
        self.triangle_wt = reg_decay(it, n, r * o.triangle_wt, o.triangle_wt)
        self.symmetry_wt = reg_decay(it, n, r * o.symmetry_wt, o.symmetry_wt)
        self.cycle_loss_wt = reg_decay(it, n, r * o.cycle_loss_wt, o.cycle_loss_wt)
        self.cycle_loss_pt_wt = reg_decay(it, n, r * o.cycle_loss_pretrain_wt, o.cycle_loss_pretrain_wt)
        self.match_wt = reg_decay(it, n, o.match_wt, r * o.match_wt)
        self.imatch_wt = reg_decay(it, n, o.imatch_wt, r * o.imatch_wt)
