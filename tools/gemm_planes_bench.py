"""tools/gemm_planes_bench.py -- the four linear layers of one ViT-S/8 block at B=32 (M = 32800 tokens) on the split main loop, A operand
split in registers (round 3, CORE 1) against A pre-split as bf16 planes (round 4, CORE 3), with the epilogue variants the block uses:
qkv (planes in, fp32 out), proj (fp32 in, fp32 + planes out), fc1 (planes in, planes out only), fc2 (planes in, fp32 + planes out).
Isolated launches on an idle device; errors vs float64 on the first 2048 rows."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import dino  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M = int(sys.argv[1]) if len(sys.argv) > 1 else 32800
g = torch.Generator(device="cuda").manual_seed(0)
tot = {"regs": 0.0, "planes": 0.0}
for name, K, N, epi, a_pl, out_pl, fp32_out in (("qkv  LN", 384, 1152, dino.GEMM_LN, True, False, True),
                                                ("proj +res", 384, 384, dino.GEMM_BIAS_RESIDUAL, False, True, True),
                                                ("fc1  LN+GELU", 384, 1536, dino.GEMM_LN_GELU, True, True, False),
                                                ("fc2  +res", 1536, 384, dino.GEMM_BIAS_RESIDUAL, True, True, True)):
    A = torch.randn(M, K, device="cuda", generator=g) * 1.3 + 0.2
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    v0 = torch.randn(N, device="cuda", generator=g) * 0.1
    v1 = torch.randn(N, device="cuda", generator=g) * 0.1
    st = torch.rand(M, 2, device="cuda", generator=g) + 0.5
    resid = torch.randn(M, N, device="cuda", generator=g)
    w3, w3t, a3 = dino.split_weight(W), dino.split_tiled(W), dino.split_tiled(A)
    o3 = dino.TiledPlanes(M, N, "cuda")
    out = torch.empty(M, N, device="cuda")
    ln = epi in (dino.GEMM_LN, dino.GEMM_LN_GELU)
    kw = dict(vec1=v1 if ln else None, rowstat=st if ln else None, resid=resid if epi == dino.GEMM_BIAS_RESIDUAL else None, epilogue=epi,
              mode="split")
    regs = lambda: dino.vit_linear(A, W, v0, out=out, w_split=w3, **kw)
    planes = lambda: dino.vit_linear(None if a_pl else A, W, v0, out=out if fp32_out else None, a_planes=a3 if a_pl else None,
                                     out_planes=o3 if out_pl else None, fp32_out=fp32_out, w_split=w3t if a_pl else w3, **kw)
    regs()
    ref_out = out.clone()
    out.zero_()
    planes()
    a64, w64 = A[:2048].double(), W.double()
    z = a64 @ w64.t()
    if ln:
        exact = st[:2048, 1:2].double() * (z - st[:2048, 0:1].double() * v0.double()) + v1.double()
        if epi == dino.GEMM_LN_GELU:
            exact = F.gelu(exact)
    else:
        exact = z + v0.double() + resid[:2048].double()
    scale = exact.abs().max().item()
    got = (o3.untile()[:, :2048].double().sum(0) if not fp32_out else out[:2048].double())
    err_p = (got - exact).abs().max().item() / scale
    err_r = (ref_out[:2048].double() - exact).abs().max().item() / scale
    consistent = bool(torch.equal(o3.untile().float().double().sum(0), out.double())) if (out_pl and fp32_out) else None
    t_r, t_p = timeit(regs), timeit(planes)
    fl = 2.0 * M * N * K
    tot["regs"] += t_r
    tot["planes"] += t_p
    print("%-14s K=%4d N=%4d | A split in registers %.3f ms = %5.1f TF/s | planes %.3f ms = %5.1f TF/s | err/scale regs %.2e planes %.2e | planes == fp32 out: %s"
          % (name, K, N, t_r, fl / t_r / 1e9, t_p, fl / t_p / 1e9, err_r, err_p, consistent))
fl_block = 2.0 * M * 384 * (1152 + 384 + 1536 + 1536)
print("block of four: A split in registers %.3f ms = %.1f TF/s | pre-split planes %.3f ms = %.1f TF/s (fp32-equivalent flops)"
      % (tot["regs"], fl_block / tot["regs"] / 1e9, tot["planes"], fl_block / tot["planes"] / 1e9))
