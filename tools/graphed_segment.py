"""tools/graphed_segment.py -- EXPERIMENT, not product code (moved out of scp_amd/graphed.py in round 5): HIP-graph replay of an
autograd segment, forward AND backward graphs behind one autograd Function.  Measured in round 4: host enqueue 21.2 -> 8.5-10.2 ms per
step, step time unchanged (device-bound); ending the capture of a BACKWARD graph takes the interpreter down inside hipStreamEndCapture in
most process setups on this ROCm 7.0 / torch 2.10 stack (tools/graph_mlp.py reproduces it with a two-layer MLP).  Kept for whoever
revisits it; nothing in scp_amd imports it."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import capi  # noqa: E402


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, n_in, *tensors):
        g = seg.graphs
        for dst, src in zip(g["inputs"], tensors[:n_in]):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        g["fwd"].replay()
        ctx.seg = seg
        outs = tuple(o.detach() for o in g["outputs"])
        # outputs that carried no gradient in the captured function carry none here either
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i not in g["out_req"]])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        g = ctx.seg.graphs
        for dst, src in zip(g["grad_outputs"], [grads[i] for i in g["out_req"]]):
            if src is None:
                dst.zero_()
            elif dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        g["bwd"].replay()
        return (None, None) + tuple(None if t is None else t.detach() for t in g["grad_inputs"])


class GraphedSegment:
    def __init__(self, fn, params, warmup=2, name="segment"):
        self.fn, self.params, self.warmup, self.name = fn, [p for p in params], warmup, name
        self.calls = 0
        self.graphs = None
        self.key = None
        self.disabled = False

    def _signature(self, inputs):
        return tuple((tuple(t.shape), t.dtype, t.requires_grad, tuple(t.stride())) for t in inputs) + tuple(p.requires_grad for p in self.params)

    def __call__(self, *inputs):
        ok = (not self.disabled and torch.is_grad_enabled() and all(t.is_cuda for t in inputs)
              and not torch.cuda.is_current_stream_capturing())
        if not ok:
            return self.fn(*inputs)
        self.calls += 1
        if self.calls <= self.warmup:
            return self.fn(*inputs)
        key = self._signature(inputs)
        if self.graphs is None:
            self._capture(inputs)
            self.key = key
        elif key != self.key:
            return self.fn(*inputs)              # another shape (last partial batch, eval ...): eager
        return _Replay.apply(self, len(inputs), *inputs, *self.params)

    def _capture(self, inputs):
        stream = torch.cuda.current_stream()
        capi.reserve_graph_tickets(inputs[0].device)
        static_in = [t.detach().clone().requires_grad_(t.requires_grad) for t in inputs]
        pool = torch.cuda.graph_pool_handle()
        fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        capi.CAPTURING = True
        try:
            with torch.cuda.graph(fwd, pool=pool):
                try:
                    outs = self.fn(*static_in)
                except BaseException:
                    import sys
                    import traceback
                    traceback.print_exc()
                    sys.stderr.flush()
                    raise
            outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
            out_req = [i for i, o in enumerate(outs) if o.requires_grad]
            grad_outs = [torch.empty_like(outs[i]) for i in out_req]
            wrt = [t for t in static_in + self.params if t.requires_grad]
            with torch.cuda.graph(bwd, pool=pool):
                try:
                    got = torch.autograd.grad([outs[i] for i in out_req], wrt, grad_outs, allow_unused=True)
                except BaseException:
                    import sys
                    import traceback
                    traceback.print_exc()              # ending an invalidated capture can take the process down: say why first
                    sys.stderr.flush()
                    raise
        finally:
            capi.CAPTURING = False
        it = iter(got)
        grad_inputs = [next(it) if t.requires_grad else None for t in static_in + self.params]
        self.graphs = dict(fwd=fwd, bwd=bwd, inputs=static_in, outputs=outs, out_req=out_req, grad_outputs=grad_outs,
                           grad_inputs=grad_inputs, pool=pool)
        torch.cuda.current_stream().wait_stream(stream)


