"""minimal repro: GraphedSegment around a torch MLP; argv: comma list of pre-imports / settings"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
opts = sys.argv[1].split(",") if len(sys.argv) > 1 else []
import torch
if "dist" in opts:
    import torch.distributed
if "bench" in opts:
    import bench
if "benchmark" in opts:
    torch.backends.cudnn.benchmark = True
if "tunable" in opts:
    torch.cuda.tunable.enable(True)
if "warn" in opts:
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graphed_segment import GraphedSegment
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.ReLU(), torch.nn.Linear(128, 32)).cuda()
seg = GraphedSegment(lambda x: (net(x),), list(net.parameters()), warmup=2)
for it in range(5):
    x = torch.randn(16, 64, device="cuda", requires_grad=True)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    (y,) = seg(x)
    y.sum().backward()
torch.cuda.synchronize()
print("OK", opts)
