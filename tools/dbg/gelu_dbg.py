import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "self-corr-pose_amd"))
from scp_amd import dino
for (M, K, N) in [(1, 32, 1), (1, 64, 1), (3, 32, 2), (64, 32, 128), (1, 32, 1)]:
    g = torch.Generator().manual_seed(M + K + N)
    a = (torch.randn(M, K, generator=g) * 1.3 + 0.4).cuda()
    w = (torch.randn(N, K, generator=g) * 0.06).cuda()
    b = (torch.randn(N, generator=g) * 0.2).cuda()
    norm = torch.nn.LayerNorm(K, eps=1e-6).cuda()
    with torch.no_grad():
        wg, s, t = dino.fold_layernorm(norm, w, b)
        st = dino.row_mean_rstd(a, 1e-6)
        ln64 = torch.nn.functional.layer_norm(a.double(), (K,), norm.weight.double(), norm.bias.double(), 1e-6)
        ref = ln64 @ w.double().t() + b.double()
        for rep in range(3):
            y = dino.vit_linear(a, wg, s, t, st, epilogue=dino.GEMM_LN)
            z = dino.vit_linear(a, wg, s, t, st, epilogue=dino.GEMM_LN_GELU)
            print(M, K, N, "ln err %.2e" % (y.double() - ref).abs().max().item(), "gelu err %.2e" % (z.double() - torch.nn.functional.gelu(ref)).abs().max().item(),
                  "ln", y.flatten()[:3].tolist(), "gelu", z.flatten()[:3].tolist(), "ref gelu", torch.nn.functional.gelu(ref).flatten()[:3].tolist())
