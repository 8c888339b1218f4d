"""scp_amd/graphed.py -- HIP-graph replay of a static, single-stream segment of the training step, forward AND backward.

Why: the step issues ~1 100 kernel launches from Python (autograd Functions around the C ABI, ATen glue); the host needs ~21-23 ms
to enqueue them (profiles/r03_host_enqueue.txt), which is within reach of the 31 ms the device needs.  The two encoder passes
(ResNet18 trunk + U-decoder + heads; ~100 launches forward and ~330 backward EACH) and the frozen ViT have static shapes and run
on one stream each: their launch sequences are captured once into HIP graphs (hipStreamBeginCapture through torch.cuda.graph) and
replayed with one hipGraphLaunch per direction.  The streams around them -- main, ViT, rotation-cycle, texture -- stay eager, so the
step keeps its multi-stream schedule (a capture of the WHOLE step replays slower than eager on this stack: HIP's graph executor
serialises the branches, tools/graph_probe.py: 33.98 vs 31.72 ms).

`GraphedSegment(fn, params)` wraps `fn(*tensors) -> tuple of tensors`:
  * the first `warmup` calls run eagerly (MIOpen's solver search for the stem / stride-2 backward must not happen under capture);
  * the next call captures two graphs into one private memory pool: forward `outs = fn(*static_inputs)` and backward
    `torch.autograd.grad(outs, inputs + params, static_grad_outs)`;
  * every later call is an autograd Function: copy the inputs into the static buffers (skipped when the caller already passes the
    same storage), replay forward, hand out the static outputs; its backward copies the incoming gradients, replays backward and
    returns the static input / parameter gradients (AccumulateGrad adds them into the trainer's flat gradient views).
Contract for `fn`: static shapes, no host synchronisation, no host-side random draws baked into kernel arguments, every workspace
allocated through torch (the pool owns it), in-place updates only of tensors whose address is stable (BatchNorm running
statistics).  Host-side caches that decide WHETHER a kernel is launched (scp_amd.fused_conv.weight_planes) must be filled before
capture and refreshed outside (Trainer.step does).  Outputs are overwritten by the next replay: consumers read them within the
step (stream-ordered), which is how the training step uses them.  The parameter gradients a backward replay returns are static
buffers too: parameters should have their .grad pre-allocated (the Trainer's flat gradient views are), so that autograd adds into
it instead of adopting the static buffer as .grad.
Numerics: a replay launches exactly the kernels of the eager call with the same arguments -- bit-identical results (tests)."""
import torch

from . import capi


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


class GraphedInference:
    """forward-only variant for a frozen, no-grad segment (the DINO ViT + pair matching): `fn(*tensors) -> tuple of tensors`.
    The outputs are CLONED out of the static buffers (they are small: indices and matched grid points), because the look-ahead
    replays the graph for the next batch while this step's backward still reads this batch's results."""

    def __init__(self, fn, warmup=2, clone_outputs=True):
        self.fn, self.warmup, self.clone_outputs = fn, warmup, clone_outputs
        self.calls = 0
        self.graph = None
        self.key = None
        self.disabled = False

    def __call__(self, *inputs):
        if self.disabled or not all(t.is_cuda for t in inputs) or torch.cuda.is_current_stream_capturing():
            return self.fn(*inputs)
        self.calls += 1
        if self.calls <= self.warmup:
            return self.fn(*inputs)
        key = tuple((tuple(t.shape), t.dtype, tuple(t.stride())) for t in inputs)
        if self.graph is None:
            capi.reserve_graph_tickets(inputs[0].device)
            self.static_in = [t.detach().clone() for t in inputs]
            self.graph = torch.cuda.CUDAGraph()
            capi.CAPTURING = True
            try:
                with torch.no_grad(), torch.cuda.graph(self.graph):
                    outs = self.fn(*self.static_in)
            finally:
                capi.CAPTURING = False
            self.static_out = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
            self.key = key
        elif key != self.key:
            return self.fn(*inputs)
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return tuple(o.clone() for o in self.static_out) if self.clone_outputs else self.static_out
