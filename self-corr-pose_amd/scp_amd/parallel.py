"""scp_amd/parallel.py -- data-parallel gradient averaging over RCCL (xGMI) / gloo.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  The batch shards by image
(train.py:29-36, data/dataloader.py:57-64); the model is replicated; per step the trainable
gradients (~59 MB fp32) are averaged.  The reference builds a DistributedDataParallel wrapper but
calls the bare module, so its reducer never fires; north_star asks for a real all-reduce,
which is what this does.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), ring all-reduce is per-link bound, so the
gradients travel as ONE large message: the trainer flattens them into a persistent buffer (stable device
address for the collective) and `all_reduce_flat` reduces it with a single call on the current stream.
`all_reduce()` (bucketed, tolerant of parameters without a gradient on some rank) is the general form.
"""
import os

import torch
import torch.distributed as dist


class GradientAllReducer:
    def __init__(self, model, process_group=None, bucket_bytes=32 << 20):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def broadcast_parameters(self, src=0):
        """replicas start identical (DDP's init broadcast, trainer.py:70-75)"""
        for p in self.params:
            dist.broadcast(p.data, src, group=self.group)

    def all_reduce_flat(self, flat):
        """average an already flattened gradient buffer in place with ONE collective over the whole buffer (~58 MB fp32 for
        this model: xGMI ring all-reduce is per-link bound, so one large message beats several small ones).  Enqueued on the
        current stream (RCCL) -- no host synchronisation.  Ranks must pass buffers of equal length (parameters without a
        gradient on some rank: use all_reduce())."""
        if self.world == 1 or os.environ.get("SCP_DEBUG_SKIP_ALLREDUCE") == "1":
            return flat
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        return flat

    def all_reduce(self):
        if self.world == 1:
            return
        work = []
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            self._flat[i] = (flat, grads, bucket)
            work.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w, (flat, grads, bucket) in zip(work, self._flat):
            w.wait()
            flat.div_(self.world)
            off = 0
            for p, g in zip(bucket, grads):
                n = g.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
