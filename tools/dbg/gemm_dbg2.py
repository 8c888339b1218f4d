import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "self-corr-pose_amd"))
from scp_amd import dino
torch.manual_seed(0)
for (M, K, N) in [(64, 32, 128), (64, 64, 128), (64, 96, 128)]:
    a = torch.randn(M, K, device="cuda") * 1.3 + 0.4; w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    st = torch.zeros(M, 2, device="cuda"); st[:, 1] = 1.0     # mean 0, rstd 1: LN epilogue == acc + t
    s = torch.zeros(N, device="cuda")
    ref = a.double() @ w.double().t() + b.double()
    for rep in range(6):
        y = dino.vit_linear(a, w, s, b, st, epilogue=dino.GEMM_LN)
        d = (y.double() - ref).abs()
        idx = (d > 1e-3).nonzero()
        print(M, K, N, "rep", rep, "bad", idx.shape[0], end=" ")
        if idx.shape[0]:
            rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
            print("rows", rows[:40], "cols", cols[:40])
            i, j = idx[0].tolist()
            print("   e.g. y[%d,%d]=%g ref %g ; acc-only ref %g" % (i, j, y[i, j].item(), ref[i, j].item(), (ref - b.double())[i, j].item()))
        else:
            print()
