"""tools/pk_forms_probe.py -- which packed-fp32 instruction form is disturbed by bf16 MFMAs on the same SIMD?  (tools/probes/pk_forms.hip)

For every form: PASSES launches of the self-checking victim on the main stream while the side stream runs (a) nothing, (b) a register-only
loop of v_mfma_f32_32x32x2_f32, (c) the same loop of v_mfma_f32_32x32x16_bf16.  Prints the number of wrong lane-results and of passes
that saw any.  A form that is wrong only under (c) is the victim instruction."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libpk_forms.so"))
FORMS = {0: "v_pk_mul_f32 plain", 1: "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", 2: "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]",
         3: "v_pk_mul_f32 op_sel_hi:[1,0]", 4: "v_pk_mul_f32 op_sel_hi:[0,1]", 5: "v_pk_add_f32 neg_lo/neg_hi", 6: "v_pk_fma_f32 plain",
         7: "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] neg", 8: "v_pk_mov_b32 op_sel:[1,0]", 9: "scalar v_mul_f32 (control)",
         10: "v_pk_fma_f32 neg_lo/neg_hi on src0", 11: "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[1,0]",
         12: "face_setup opening (3 op_sel products, scalar writes over their high halves, packed subtract)",
         20: "form 1 on operands straight from memory", 21: "form 2 on operands straight from memory"}
PASSES = int(os.environ.get("PASSES", "40"))
dev = "cuda"
side = torch.cuda.Stream()
out = torch.zeros(4096 * 256, dtype=torch.int32, device=dev)
mem = (torch.rand(1 << 22, device=dev) + 0.5).contiguous()
aggr_out = torch.empty(2048 * 256, device=dev)


def run(form, aggr):
    wrong, passes = 0, 0
    for _ in range(PASSES):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        if aggr is not None:
            with torch.cuda.stream(side):
                for _k in range(6):
                    code = LIB.pk_aggressor(aggr, ctypes.c_void_p(aggr_out.data_ptr()), 1024, 6000, ctypes.c_void_p(side.cuda_stream))
                    assert code == 0, code
        for _k in range(3):
            out.zero_()
            code = LIB.pk_victim(form, ctypes.c_void_p(out.data_ptr()), 4096, 400, ctypes.c_void_p(mem.data_ptr()), ctypes.c_void_p(main.cuda_stream))
            assert code == 0, code
            n = int(out.sum())
            wrong += n
            passes += int(n > 0)
        main.wait_stream(side)
    return wrong, passes


forms = [int(a) for a in sys.argv[1:]] or sorted(FORMS)
print("%d passes x 3 launches of 1 M threads x 400 iterations per cell; cell = wrong lane-results / launches with any" % PASSES)
print("%-95s %14s %14s %14s" % ("form", "alone", "fp32 MFMA", "bf16 MFMA K=16"))
for f in forms:
    cells = [run(f, a) for a in (None, 1, 0)]
    print("%-95s %14s %14s %14s" % ("%2d %s" % (f, FORMS[f]), *["%d / %d" % c for c in cells]), flush=True)
