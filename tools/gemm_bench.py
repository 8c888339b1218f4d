"""tools/gemm_bench.py -- the four linear layers of one ViT-S/8 block at B=32 (M = 32800 tokens): csrc/vit_gemm.hip (fused
epilogues) against F.linear (+ the separate LayerNorm / GELU / residual kernels it replaces); numerics vs float64."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd import capi, dino  # noqa: E402

L = capi.lib()
if os.environ.get("GEMM_LIB"):          # an ablation build of csrc/vit_gemm.hip (tools/probes), timing only
    L0, L = L, ctypes.CDLL(os.environ["GEMM_LIB"])
    for f in ("scp_vit_linear", "scp_row_mean_rstd"):
        getattr(L, f).argtypes, getattr(L, f).restype = getattr(L0, f).argtypes, getattr(L0, f).restype
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


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


SPLIT = os.environ.get("SCP_VIT_GEMM", "split") == "split"      # main loop: bf16 cores on exactly split operands, or fp32 cores
_PLANES = {}


def run(A, W, v0, v1, st, resid, C, epi):
    wp = W
    if SPLIT:
        if id(W) not in _PLANES:
            _PLANES[id(W)] = (W, dino.split_weight(W))
        wp, epi = _PLANES[id(W)][1], epi | dino.GEMM_W_SPLIT3
    capi.check(L.scp_vit_linear(P(A), P(wp), P(v0), P(v1), P(st), P(resid), P(C), A.shape[0], W.shape[0], A.shape[1], epi,
                                capi.current_stream()), "vit_linear")


M = int(sys.argv[1]) if len(sys.argv) > 1 else 32800
g = torch.Generator(device="cuda").manual_seed(0)
tot_mine = tot_ref = 0.0
for name, K, N, epi in (("qkv  LN", 384, 1152, 2), ("proj +res", 384, 384, 1), ("fc1  LN+GELU", 384, 1536, 3), ("fc2  +res", 1536, 384, 1)):
    A = torch.randn(M, K, device="cuda", generator=g) * 1.3 + 0.2
    Wo = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    C = torch.empty(M, N, device="cuda")
    if epi in (2, 3):
        gamma = 1 + 0.1 * torch.randn(K, device="cuda", generator=g)
        beta = 0.1 * torch.randn(K, device="cuda", generator=g)
        W = (Wo * gamma[None]).contiguous()
        v0 = W.sum(1).contiguous()
        v1 = (Wo @ beta + b).contiguous()
        st = torch.empty(M, 2, device="cuda")
        mine = lambda: (capi.check(L.scp_row_mean_rstd(P(A), P(st), M, K, 1e-6, capi.current_stream()), "stats"),
                        run(A, W, v0, v1, st, None, C, epi))
        if epi == 2:
            ref = lambda: F.linear(F.layer_norm(A, (K,), gamma, beta, 1e-6), Wo, b)
        else:
            ref = lambda: F.gelu(F.linear(F.layer_norm(A, (K,), gamma, beta, 1e-6), Wo, b))
        mine()
        ln64 = F.layer_norm(A[:4096].double(), (K,), gamma.double(), beta.double(), 1e-6)
        exact = ln64 @ Wo.double().t() + b.double()
        if epi == 3:
            exact = F.gelu(exact)
        got, r32 = C[:4096].double(), ref()[:4096].double()
    else:
        resid = torch.randn(M, N, device="cuda", generator=g)
        W = Wo
        mine = lambda: run(A, W, b, None, None, resid, C, epi)
        ref = lambda: resid + F.linear(A, Wo, b)
        mine()
        exact = resid[:4096].double() + A[:4096].double() @ Wo.double().t() + b.double()
        got, r32 = C[:4096].double(), ref()[:4096].double()
    scale = exact.abs().max().item()
    t_m, t_r = timeit(mine), timeit(ref)
    only = timeit(lambda: run(A, W, v0 if epi in (2, 3) else b, v1 if epi in (2, 3) else None, st if epi in (2, 3) else None,
                              None if epi in (2, 3) else resid, C, epi))
    fl = 2.0 * M * N * K
    tot_mine += t_m
    tot_ref += t_r
    print("%-14s K=%4d N=%4d | own %.3f ms (GEMM alone %.3f ms = %5.1f TF/s) | torch composition %.3f ms | err own %.2e torch %.2e (of scale %.2f)"
          % (name, K, N, t_m, only, fl / only / 1e9, t_r, (got - exact).abs().max().item() / scale,
             (r32 - exact).abs().max().item() / scale, scale))
print("block total: own %.3f ms, torch composition %.3f ms" % (tot_mine, tot_ref))
