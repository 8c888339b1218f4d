"""tools/bf16x6_probe.py -- evidence for DESIGN.md section 7 ("next"): an fp32 GEMM evaluated as SIX bf16
products with fp32 accumulation (a = hi + mid + lo, each bf16; products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi,
mid.mid; K-concatenated into one bf16 -> fp32 library GEMM) has the SAME error as the native fp32 GEMM
(measured against fp64) and runs 1.3-1.5x faster on the MI355X matrix cores for the ViT's K=384 layers.
NOT used by the product or the bench (which stay plain fp32)."""
import torch, time
torch.manual_seed(0)
M = 32800
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(True); e1=torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
def split3(x):
    hi = x.to(torch.bfloat16); r1 = x - hi.float(); mid = r1.to(torch.bfloat16); r2 = r1 - mid.float(); lo = r2.to(torch.bfloat16)
    return hi, mid, lo
for (K, N) in ((384, 1152), (384, 1536), (1536, 384), (384, 384)):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
    ah, am, al = split3(a); wh, wm, wl = split3(w)
    A6 = torch.cat([ah, ah, am, ah, al, am], 1).contiguous()
    B6 = torch.cat([wh, wm, wh, wl, wh, wm], 1).t().contiguous()      # [6K, N]
    ref64 = (a.double() @ w.double().t())
    f32 = a @ w.t()
    try:
        out = torch.mm(A6, B6, out_dtype=torch.float32)
    except Exception as e:
        print("out_dtype unsupported:", repr(e)[:200]); break
    scale = ref64.abs().mean()
    print("K=%4d N=%4d  fp32 gemm %.3f ms | bf16x6 gemm %.3f ms (+split %.3f ms) | err fp32 %.2e  bf16x6 %.2e (mean abs / mean |ref|)" % (
        K, N, t(lambda: a @ w.t()), t(lambda: torch.mm(A6, B6, out_dtype=torch.float32)),
        t(lambda: torch.cat([x for x in (lambda h,m,l: (h,h,m,h,l,m))(*split3(a))], 1)),
        ((f32.double()-ref64).abs().mean()/scale).item(), ((out.double()-ref64).abs().mean()/scale).item()))
