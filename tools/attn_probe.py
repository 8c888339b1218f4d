"""tools/attn_probe.py -- interleaved timing of the attention kernel under its debug switches (results are WRONG
under most of them; this only prices the pieces): 2 = no vmcnt wait, 8 = no LDS-DMA, 16 = no per-tile barrier,
64 = plain (non XCD-aware) workgroup order."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]]
exec(open(os.path.join(ROOT, "tools", "attn_bench.py")).read().split('if "--ablate"')[0])
from scp_amd import capi  # noqa: E402
lib = capi.lib()
for flags in (0, 16, 0, 2, 0, 18, 0, 26, 0, 8, 0, 64, 0):
    lib.scpdbg_set_attn_flags(flags)
    print("flags=%2d %.3f ms" % (flags, timeit(lambda: fused_attention(qkv, B, N, H, 64, 0.125), it=50)))
lib.scpdbg_set_attn_flags(0)
