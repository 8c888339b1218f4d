import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "self-corr-pose_amd"), os.path.join(ROOT, "tests")]
import scp_amd.dino as dino
from scp_amd.flags import Options
from scp_amd.model import MeshNet
from scp_amd import synthetic
dino.ALLOW_RANDOM_INIT = True
opts = Options("laptop_wild6d", batch_size=1, repeat=2, train=True)
torch.manual_seed(0)
m = MeshNet(opts, prior=synthetic.bottle_like(2)).cuda().train()
m.rotation_angle = 90.0
for jit in (True, False):
    if not jit:
        m.encoder.random_jitter = torch.nn.Identity()
    for p in m.parameters():
        p.grad = None
    data = synthetic.make_batch(1, 2, 256, seed=3, device="cuda")
    total, aux = m(data)
    total.mean().backward()
    g = m.encoder.backbone.resnet.conv1.weight.grad
    print("jitter" if jit else "identity", "conv1 grad:", None if g is None else (float(g.norm()), bool(torch.isfinite(g).all())), "total", float(total.mean()))
