import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "self-corr-pose_amd"))
from scp_amd import dino
torch.manual_seed(0)
for (M, K, N) in [(64, 32, 128), (64, 384, 128), (1, 32, 1), (300, 32, 384), (2048, 384, 384), (3075, 384, 1152)]:
    a = torch.randn(M, K, device="cuda") * 1.3 + 0.4; w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    norm = torch.nn.LayerNorm(K, eps=1e-6).cuda()
    with torch.no_grad():
        wg, s, t = dino.fold_layernorm(norm, w, b)
        st = dino.row_mean_rstd(a, 1e-6)
        ln64 = torch.nn.functional.layer_norm(a.double(), (K,), norm.weight.double(), norm.bias.double(), 1e-6)
        ref = ln64 @ w.double().t() + b.double()
        for epi, name, rf in ((dino.GEMM_LN, "ln", ref), (dino.GEMM_LN_GELU, "gelu", torch.nn.functional.gelu(ref))):
            bad = 0; worst = 0.0; info = ""
            for rep in range(60):
                y = dino.vit_linear(a, wg, s, t, st, epilogue=epi)
                d = (y.double() - rf).abs()
                e = d.max().item()
                if not e < 1e-3:
                    bad += 1
                    if not info:
                        idx = (d > 1e-3).nonzero()
                        info = "first bad: %d elems, rows %d..%d cols %d..%d" % (idx.shape[0], idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max())
                else:
                    worst = max(worst, e)
            print(M, K, N, name, "bad %d/60 worst-good %.3e" % (bad, worst), info, flush=True)
