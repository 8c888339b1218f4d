"""scp_amd/parallel.py -- data-parallel gradient averaging over RCCL (xGMI) / gloo, overlapped with backward.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  The batch shards by image
(train.py:29-36, data/dataloader.py:57-64); the model is replicated; per step the trainable gradients
(~58 MB fp32) are averaged.  The reference builds a DistributedDataParallel wrapper (model/trainer.py:70-75) but
calls the bare module, so its reducer never fires; north_star asks for a real all-reduce.

FlatGradients (used by the Trainer at every world size):
  * ONE persistent flat fp32 buffer holds every trainable gradient; each `p.grad` is a VIEW into it, so autograd
    accumulates straight into the buffer (no per-step cat / copy-back passes) and the collectives always see the
    same device addresses (no re-registration inside RCCL).
  * the buffer is laid out in REVERSE parameter-registration order (heads and decoder first, ResNet stem last =
    roughly the order backward produces them) and cut into ~25 MB buckets, each a contiguous slice.
  * a post-accumulate-grad hook per parameter counts a bucket down; the moment its last gradient has been
    accumulated the bucket's all-reduce is enqueued (async_op) on a dedicated communication stream that waits
    on the producing stream -- so buckets travel over xGMI while the rest of backward is still computing.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU) and ring all-reduce is per-link bound: few large
    messages (3 buckets here), not many small ones.
  * finish(): launches whatever did not fire (parameters without a gradient this step contribute zeros on
    every rank, so the collective sequence is identical everywhere), makes the compute stream wait for the
    collectives and returns the flat buffer; the 1/world scale is folded into the caller's clip/NaN pass.
"""
import torch
import torch.distributed as dist

from . import streams


class FlatGradients:
    def __init__(self, params, process_group=None, bucket_bytes=25 << 20, distributed=None, force_collectives=False):
        """force_collectives: run the hook / communication-stream / all-reduce path even in a 1-rank group (lets the RCCL path be
        exercised on a single GPU, tests/test_parallel.py)"""
        self.group = process_group
        self.distributed = dist.is_initialized() if distributed is None else distributed
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.active = self.world > 1 or (force_collectives and self.distributed)
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        order = list(reversed(self.params))
        self.flat = torch.zeros(sum(p.numel() for p in order), dtype=dt, device=dev)
        self.views, self.span, self.bucket_of = {}, {}, {}
        self.buckets = []                      # (lo, hi, [params])
        off, lo, cur = 0, 0, []
        for p in order:
            n = p.numel()
            if cur and (off + n - lo) * p.element_size() > bucket_bytes:
                self.buckets.append((lo, off, cur))
                lo, cur = off, []
            self.views[id(p)] = self._view_like(p, off)
            self.span[id(p)] = (off, n)
            self.bucket_of[id(p)] = len(self.buckets)
            cur.append(p)
            off += n
        self.buckets.append((lo, off, cur))
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._armed = False
        self._prepared = False
        self._silent, self._fired, self._poisoned = set(), set(), set()
        # which parameters received a gradient ON ANY RANK (decides whose .grad ends as None: see finish()).  Established by one
        # MAX all-reduce + host read in the first step (and after reset_static_graph()), then only VERIFIED on the device every step
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._global_used = None               # host: bool list per parameter, identical on every rank
        self._global_used_dev = None
        self._used_mismatch = None             # device flag: some rank's used set differed from the established one
        self.launched_in_backward = 0          # buckets whose collective was enqueued from a hook (test / log)
        # SCP_STREAMS=serial (scp_amd/streams.py): no communication stream and no launches from the gradient hooks -- the buckets go
        # out from finish(), after backward, so that the reduction kernels never share a compute unit with the backward's bf16-MFMA kernels
        self.overlap = dev.type != "cuda" or streams.overlap()
        self.comm_stream = streams.side_stream(dev) if dev.type == "cuda" and self.active and self.overlap else None
        # hooks at every world size: with one rank they only record which parameters received a gradient (finish() needs that
        # to leave the unused ones at grad = None)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _view_like(self, p, off):
        """a view of flat[off : off + p.numel()] with p's sizes AND strides: the fused optimizer requires gradients in the
        parameter's own layout (the encoder's convolution weights are channels_last)"""
        dims = sorted(range(p.dim()), key=lambda d: p.stride(d), reverse=True)
        expect, dense = 1, True
        for d in reversed(dims):
            dense &= p.size(d) == 1 or p.stride(d) == expect
            expect *= p.size(d)
        assert dense, "parameter is neither contiguous nor a dense permutation"
        return torch.as_strided(self.flat, p.size(), p.stride(), off)

    # ------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, module=None, src=0):
        """replicas start identical (DDP's init broadcast, trainer.py:70-75); with `module`, its buffers
        (BatchNorm running statistics) too"""
        if self.world == 1:
            return
        tensors = [p.data for p in (module.parameters() if module is not None else self.params)]
        if module is not None:
            tensors += [b.data for b in module.buffers() if b.is_floating_point() or b.dtype in (torch.int64, torch.int32)]
        for t in tensors:
            dist.broadcast(t, src, group=self.group)

    def averaged_running_stats(self, module):
        """{buffer name: mean over ranks} for the BatchNorm running_mean / running_var buffers of `module` (each rank normalises
        with its own 32 images, like the reference without SyncBatchNorm; a checkpoint should not depend on which rank
        writes it).  COLLECTIVE: every rank must call it.  The live buffers are not touched and nothing else is reduced
        (constant buffers and integer counters are identical on every rank by construction)."""
        out = {}
        if self.world == 1:
            return out
        for name, b in module.named_buffers():
            if b.is_floating_point() and (name.endswith("running_mean") or name.endswith("running_var")):
                avg = b.detach().clone()
                dist.all_reduce(avg, op=dist.ReduceOp.SUM, group=self.group)
                out[name] = avg.div_(self.world)
        return out

    # ------------------------------------------------------------------------------------------------
    def prepare(self):
        """before forward/backward: zero the buffer and point every p.grad at its view (replaces optimizer.zero_grad)"""
        self.flat.zero_()
        for p in self.params:
            v = self.views[id(p)]
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        # parameters that produced no gradient in the previous step (heads that a flag switches off, ...) are not waited
        # for: their bucket goes out as soon as the others are in (DDP's static-graph assumption; if the set changes the
        # bucket is simply launched by finish() instead -- the result is the same, only the overlap is lost)
        self._counted = [[id(p) for p in b[2] if id(p) not in self._silent] for b in self.buckets]
        self._pending = [len(c) for c in self._counted]
        self._work = [None] * len(self.buckets)
        self._fired = set()
        self._poisoned = set()
        self.launched_in_backward = 0
        self._armed = self.overlap
        self._prepared = True

    def _launch(self, i):
        lo, hi, _ = self.buckets[i]
        seg = self.flat[lo:hi]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._work[i] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._work[i] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        v = self.views[id(p)]
        if self._prepared and p.grad is not None and p.grad.data_ptr() != v.data_ptr():     # autograd replaced the view (first-touch steal)
            v.copy_(p.grad)
            p.grad = v
        if not self._prepared:
            return
        i = self.bucket_of[id(p)]
        first = id(p) not in self._fired
        self._fired.add(id(p))                            # (also without overlap: finish() tells used from unused parameters by it)
        if not self._armed or not first or not self.active:
            return
        if id(p) in self._silent:
            # a parameter that had no gradient in the previous step produces one now (an iteration- or flag-dependent
            # branch): it was left out of its bucket's count-down.
            if self._work[i] is not None:
                # its bucket is already being reduced: the accumulation that just ran raced with the in-place all-reduce
                # on the communication stream and the other ranks summed a stale value -- not recoverable
                raise RuntimeError(
                    "FlatGradients: a parameter of shape %s received its first gradient after its bucket had been "
                    "all-reduced (it had none in the previous step); call reset_static_graph() before a step that changes "
                    "the set of used parameters" % (tuple(p.shape),))
            # not launched yet: only finish() may launch this bucket now (and, buckets going out in index order, every
            # later one), after all of backward
            self._poisoned.add(i)
            return
        if self._work[i] is not None:
            return
        self._pending[i] -= 1
        # buckets are launched strictly in index order (here or in finish()), so every rank issues the same
        # collective sequence whatever the timing of its hooks
        while True:
            nxt = next((j for j in range(len(self.buckets)) if self._work[j] is None), None)
            if nxt is None or self._pending[nxt] != 0 or nxt in self._poisoned:
                break
            self._launch(nxt)
            self.launched_in_backward += 1

    def reset_static_graph(self):
        """forget which parameters were unused in the previous step: every bucket waits for all of its parameters again, and the
        set of parameters used on any rank is established afresh by the next finish() (one host read).  COLLECTIVE in effect: call
        it on every rank before a step whose set of used parameters differs from the last one's."""
        self._silent = set()
        self._global_used = self._global_used_dev = self._used_mismatch = None

    def check_static_graph(self):
        """host check (one device read) that no rank's set of used parameters has departed from the established one since the
        last check; Trainer.train() calls it where it reads the losses back anyway.  Raises on every rank alike."""
        if self._used_mismatch is not None and bool(self._used_mismatch.item()):
            raise RuntimeError("FlatGradients: the set of parameters that receive a gradient changed on some rank; call "
                               "reset_static_graph() on every rank before such a step")

    def _global_unused(self, local_unused):
        """the parameters that received no gradient on ANY rank this step (the reference's DDP all-reduces its used-parameter
        bitmap for the same purpose): a parameter used on some ranks only must be stepped by ALL replicas with the averaged
        gradient, or they diverge.  One tiny MAX all-reduce per step; its result is read on the host only when the set is being
        established, afterwards it is compared on the device with the established one (check_static_graph() reads that flag)."""
        if not self.active:
            return local_unused
        key = tuple(self._index[id(p)] for p in local_unused)
        cache = self.__dict__.setdefault("_local_mask_cache", {})
        if key not in cache:                                  # built once per distinct local set (a host-to-device copy)
            m = torch.ones(len(self.params), dtype=torch.int32)
            if key:
                m[list(key)] = 0
            cache[key] = m.to(self.flat.device)
        local = cache[key].clone()
        dist.all_reduce(local, op=dist.ReduceOp.MAX, group=self.group)
        if self._global_used is None:
            self._global_used = [bool(v) for v in local.cpu().tolist()]
            self._global_used_dev = local
            self._used_mismatch = torch.zeros((), dtype=torch.bool, device=self.flat.device)
        else:
            self._used_mismatch = self._used_mismatch | (local != self._global_used_dev).any()
        return [p for p, used in zip(self.params, self._global_used) if not used]

    def finish(self, keep_unused_none=False):
        """after backward: adopt gradients that were assigned around the views, launch the buckets that have not fired,
        wait for the collectives.  Returns the flat buffer holding the SUM over ranks (divide by .world).
        keep_unused_none: parameters that received no gradient this step end with `p.grad = None` (what
        zero_grad(set_to_none=True) leaves in the reference, so AdamW skips them: no weight decay, no moment update) instead
        of a zero view; prepare() points them at their views again."""
        self._armed = False
        unused = []
        for p in self.params:
            v = self.views[id(p)]
            if p.grad is None:
                if not self._prepared:
                    v.zero_()                            # prepare() was skipped: the span still holds the previous step
                p.grad = v                               # no gradient this step: zeros
                unused.append(p)
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
            elif id(p) not in self._fired:
                unused.append(p)                         # still the zero view prepare() installed
        self._prepared = False
        if self.active:
            for i in range(len(self.buckets)):
                if self._work[i] is None:
                    self._launch(i)
            self._silent = {id(p) for p in self.params} - self._fired
            for w in self._work:
                w.wait()                                  # NCCL: the current stream waits; gloo: blocks
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        if keep_unused_none:
            for p in self._global_unused(unused):
                p.grad = None
        return self.flat


# name kept for callers of round 1
GradientAllReducer = FlatGradients
