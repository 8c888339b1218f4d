import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "self-corr-pose_amd"))
from scp_amd import capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
from scp_amd import dino
torch.manual_seed(0)
tot_bad = 0
for (M, K, N) in [(64, 32, 128), (300, 32, 384), (1, 32, 1), (64, 64, 128)]:
    a = torch.randn(M, K, device="cuda") * 1.3 + 0.4; w = torch.randn(N, K, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    norm = torch.nn.LayerNorm(K, eps=1e-6).cuda()
    with torch.no_grad():
        wg, s, t = dino.fold_layernorm(norm, w, b)
        st = dino.row_mean_rstd(a, 1e-6)
        ln64 = torch.nn.functional.layer_norm(a.double(), (K,), norm.weight.double(), norm.bias.double(), 1e-6)
        ref = ln64 @ w.double().t() + b.double()
        for epi, name, rf in ((dino.GEMM_LN, "ln", ref), (dino.GEMM_LN_GELU, "gelu", torch.nn.functional.gelu(ref)), (dino.GEMM_BIAS, "bias", None)):
            if rf is None:
                rf = a.double() @ w.double().t() + b.double()
            bad = 0
            for rep in range(40):
                y = dino.vit_linear(a, wg, s, t, st, epilogue=epi) if epi != dino.GEMM_BIAS else dino.vit_linear(a, w, b)
                if not (y.double() - rf).abs().max().item() < 1e-3:
                    bad += 1
            tot_bad += bad
            print("  %s %s: bad %d/40" % ((M, K, N), name, bad))
print(os.path.basename(capi.LIB_PATH), "TOTAL BAD", tot_bad)
