import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "self-corr-pose_amd"))
from scp_amd import dino
torch.manual_seed(0)
M, K, N = 64, 32, 128
a = torch.randn(M, K, device="cuda") * 1.3 + 0.4; w = torch.randn(N, K, device="cuda") * 0.1
for rstd, tval, sval, mean in [(1.0, 0.0, 0.0, 0.0), (2.0, 0.0, 0.0, 0.0), (1.0, 100.0, 0.0, 0.0), (1.0, 0.0, 1.0, 7.0)]:
    st = torch.zeros(M, 2, device="cuda"); st[:, 0] = mean; st[:, 1] = rstd
    s = torch.full((N,), sval, device="cuda"); t = torch.full((N,), tval, device="cuda")
    acc = a.double() @ w.double().t()
    ref = rstd * (acc - mean * sval) + tval
    for rep in range(4):
        y = dino.vit_linear(a, w, s, t, st, epilogue=dino.GEMM_LN)
        print("rstd %g t %g s %g mean %g: y[0,0]=%r ref %g | y[4,0]=%r ref %g | y[0,1]=%r" % (rstd, tval, sval, mean, y[0, 0].item(), ref[0, 0].item(), y[4, 0].item(), ref[4, 0].item(), y[0, 1].item()))
