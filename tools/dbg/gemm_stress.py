import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "self-corr-pose_amd"))
from scp_amd import dino
torch.manual_seed(0)
for (M, K, N) in [(64, 32, 128), (64, 64, 128), (64, 384, 128), (1, 32, 1), (300, 32, 384), (2048, 384, 384), (32800, 384, 1152), (3075, 384, 1152)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    ref = (a.double() @ w.double().t() + b.double())
    bad = 0; worst = 0.0
    for rep in range(60):
        y = dino.vit_linear(a, w, b)
        e = (y.double() - ref).abs().max().item()
        worst = max(worst, e)
        if not e < 1e-3:
            bad += 1
        if rep % 3 == 0:
            torch.cuda.synchronize()
    print(M, K, N, "bad %d/60 worst %.3e" % (bad, worst), flush=True)
