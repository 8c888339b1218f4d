"""tools/aten_probe.py -- isolated timings of the stock ATen ops of the step that looked slow in the in-step census
(contention inflates in-step durations; this separates the genuinely slow kernels from the merely squeezed ones)."""
import torch
import torch.nn.functional as F

dev = "cuda"


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


img = torch.rand(32, 3, 256, 256, device=dev)
mask = torch.rand(32, 1, 256, 256, device=dev)
theta = torch.tensor([[0.8, -0.6, 0.0], [0.6, 0.8, 0.0]], device=dev)[None].expand(32, -1, -1)
grid = F.affine_grid(theta, (32, 1, 256, 256), align_corners=False)
print("grid_sample bilinear [32,3,256,256]: %.1f us" % t(lambda: F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)))
print("grid_sample nearest  [32,1,256,256]: %.1f us" % t(lambda: F.grid_sample(mask, grid, mode="nearest", padding_mode="zeros", align_corners=False)))
print("affine_grid: %.1f us" % t(lambda: F.affine_grid(theta, (32, 1, 256, 256), align_corners=False)))
x = torch.randn(32, 64, 128, 128, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = F.max_pool2d(x, 3, 2, 1)
g = torch.randn_like(y)
print("max_pool fwd NHWC [32,64,128,128]: %.1f us" % t(lambda: F.max_pool2d(x, 3, 2, 1)))
print("max_pool bwd: %.1f us" % t(lambda: torch.autograd.grad(y, x, g, retain_graph=True)))
u = torch.randn(32, 128, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
print("upsample fwd NHWC [32,128,32,32]->64: %.1f us" % t(lambda: F.interpolate(u, (64, 64), mode="bilinear", align_corners=False)))
h = torch.randn(32, 1025, 1536, device=dev)
print("gelu [32,1025,1536]: %.1f us" % t(lambda: F.gelu(h)))
a = torch.randn(32, 3, 256, 256, device=dev)
print("elementwise sub [32,3,256,256]: %.1f us" % t(lambda: a - a))
s = torch.randn(32, 256, 256, device=dev)
print("sum((1,2)) [32,256,256]: %.1f us" % t(lambda: s.sum((1, 2))))
idx = torch.randint(0, 1024, (64, 200), device=dev)
pc = torch.randn(64, 1024, 642, device=dev, requires_grad=True)
sel = torch.gather(pc, 1, idx[:, :, None].expand(-1, -1, 642))
gs = torch.randn_like(sel)
print("gather rows [64,1024,642] -> 200 (fwd): %.1f us" % t(lambda: torch.gather(pc, 1, idx[:, :, None].expand(-1, -1, 642))))
print("gather rows backward: %.1f us" % t(lambda: torch.autograd.grad(sel, pc, gs, retain_graph=True)))
p = torch.randn(32, 4096, 642, device=dev, requires_grad=True)
q = (p.reshape(32, 32, 2, 32, 2, 642).sum((2, 4)) * 0.25)
gq = torch.randn_like(q)
print("pool2x2 fwd: %.1f us  bwd: %.1f us" % (t(lambda: p.reshape(32, 32, 2, 32, 2, 642).sum((2, 4)) * 0.25), t(lambda: torch.autograd.grad(q, p, gq, retain_graph=True))))
