"""scp_amd/dino.py -- frozen DINO ViT-S/8 key-feature extractor.

Behaviour restated from model/module/network/dino.py (DINO :9-108: block-9 KEYS of a ViT-S/8,
cls token dropped, channel = head*64 + d, output [n,384,32,32]; raw keys, not normalised; input not
ImageNet-normalised -- SURVEY F6) and third-party/zsp/zsp/method/vision_transformer_flexible.py
(PatchEmbed :134-149, interpolate_pos_encoding :192-212, prepare_tokens :214-225, Block :108-132,
Attention :73-101, Mlp :54-70, get_specific_tokens :249-262, vit_small :283-287).

MI355X-first differences, all output-preserving:
  * only what the output depends on is computed: blocks 0-8 in full, then norm1 and the K slice of
    block 9's qkv projection (the reference runs 12 blocks and materialises q,k,v and the
    [n,6,1025,1025] attention of blocks 9 and 11) -- SURVEY F5;
  * attention never materialises the score matrix (fused kernel), the bicubic-interpolated
    positional embedding is computed once per input size and cached;
  * every linear layer runs on the hand-written fp32-MFMA GEMM of csrc/vit_gemm.hip with its neighbours fused in:
    LayerNorm folded into the qkv / fc1 GEMMs (frozen weights: gamma o W precomputed, per-row mean / rstd applied in the
    epilogue), erf-GELU in fc1's epilogue, bias + residual add in proj's and fc2's (in place on the residual stream);
  * callers pass each unique image once; there is no empty_cache()/chunking.
The module tree and parameter names equal the DINO checkpoint's, so
`pretrain/dino_deitsmall8_pretrain.pth` loads with strict=True and MeshNet's state_dict keys are
`pretrain_corr_net.net.model.*` as in the reference.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# the reference cannot be constructed without the checkpoint file (dino.py:40-44); synthetic-weight
# runs (bench, tests) flip this on explicitly
ALLOW_RANDOM_INIT = False
# BASELINE configs[4] (mixed bf16): the six linear layers of every block run on the bf16 matrix cores (fp32
# accumulate); residual stream, LayerNorm, attention scores / softmax / P.V stay fp32.  Set from opts.mixed_bf16.
MIXED_BF16 = False
# under MIXED_BF16: the ViT through the build's own GEMM / attention kernels with bf16-rounded operands ("1"), or through
# hipBLASLt bf16 GEMMs + csrc/vit_attn_bf16.hip with bf16 activations ("0")
MIXED_OWN_KERNELS = os.environ.get("SCP_MIXED_OWN_KERNELS", "1") == "1"


def _linear(layer_or_w, x, bias=None):
    """Linear with fp32 output; under MIXED_BF16 the GEMM itself takes bf16 operands"""
    w, b = (layer_or_w.weight, layer_or_w.bias) if isinstance(layer_or_w, nn.Linear) else (layer_or_w, bias)
    if MIXED_BF16 and x.is_cuda:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return F.linear(x, w, b).float()
    return F.linear(x, w, b)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        if MIXED_BF16 and x.is_cuda:      # keep the hidden activation in bf16 between the two GEMMs (no fp32 round trip)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.fc2(F.gelu(self.fc1(x))).float()
        return _linear(self.fc2, F.gelu(_linear(self.fc1, x)))


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        if MIXED_BF16 and x.is_cuda:   # qkv GEMM -> bf16 attention -> proj GEMM without leaving bf16
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = fused_attention_bf16(self.qkv(x), b, n, self.num_heads, c // self.num_heads, self.scale)
                return self.proj(y).float()
        y = fused_attention(_linear(self.qkv, x), b, n, self.num_heads, c // self.num_heads, self.scale)   # [b,n,c]
        return _linear(self.proj, y)

    def keys(self, x):
        """only the K third of the qkv projection: [b, heads, n, d]"""
        b, n, c = x.shape
        k = _linear(self.qkv.weight[c:2 * c], x, self.qkv.bias[c:2 * c])
        return k.reshape(b, n, self.num_heads, c // self.num_heads).permute(0, 2, 1, 3)


# "split": Q K^T and P V on the bf16 matrix cores with exactly split operands (csrc/vit_attn_split.hip; fp32-accurate);
# "fp32": csrc/vit_attn.hip on the fp32 matrix cores.  SCP_VIT_ATTN=fp32 in the environment selects the latter.
ATTN_MODE = os.environ.get("SCP_VIT_ATTN", "split")


def attn_mode():
    """the attention kernel of this call: BASELINE configs[4] ("mixed bf16") rounds the operands to bf16, one product"""
    return "bf16" if MIXED_BF16 else ATTN_MODE


def gemm_mode():
    return "bf16" if MIXED_BF16 else GEMM_MODE


def attention_workspace(b, n, heads, device):
    """the operand-plane workspace of the split attention (scp_vit_attention_split_workspace bytes); the qkv projection can write the
    Q / K planes into it (vit_linear(..., qk_planes=...)) and hand it to fused_attention(presplit=...)"""
    from . import capi
    return torch.empty(capi.lib().scp_vit_attention_split_workspace(b, n, heads), dtype=torch.uint8, device=device)


def fused_attention(qkv, b, n, heads, head_dim, scale, q_rows=None, q_count=None, mode=None, presplit=None, out_planes=None):
    """HIP flash-style attention on the matrix cores (csrc/vit_attn_split.hip / csrc/vit_attn.hip, see ATTN_MODE): qkv
    [b,n,3*heads*head_dim] as produced by the qkv Linear -> [b, n, heads*head_dim].  Forward only (the DINO ViT is frozen and
    always evaluated under no_grad); GPU tensors only, no CPU fallback.
    Query selection (scp_vit_attention_forward_rows): q_rows [b,n] int32 = token index of query slot j per image, q_count [b]
    int32 = number of slots; only those tokens' outputs are produced (at their own rows; the other rows stay uninitialised),
    keys and values are always all n tokens.
    `presplit` (split mode): attention_workspace(...) into which the qkv projection already wrote the Q / K planes.  `out_planes` (with
    presplit): TiledPlanes [b*n, heads*head_dim] that receive the result as the proj GEMM's pre-split A operand; the fp32 result is then
    not produced at all and the planes are returned."""
    from . import capi
    if torch.is_grad_enabled() and qkv.requires_grad:
        raise RuntimeError("scp_amd.dino.fused_attention is forward-only (frozen ViT)")
    qkv = qkv.contiguous()
    if out_planes is not None:
        if presplit is None or not (isinstance(out_planes, TiledPlanes) and (out_planes.rows, out_planes.cols) == (b * n, heads * head_dim)):
            raise RuntimeError("fused_attention: out_planes must be TiledPlanes [b*n, heads*head_dim] and goes with presplit")
        out = None
    else:
        out = torch.empty(b, n, heads * head_dim, dtype=torch.float32, device=qkv.device)
    mode = attn_mode() if mode is None else mode
    if mode not in ("split", "fp32", "bf16"):
        raise RuntimeError("fused_attention: unknown mode %r" % (mode,))
    if q_rows is not None:
        for t, name, numel in ((q_rows, "q_rows", b * n), (q_count, "q_count", b)):
            if not (t is not None and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous() and t.numel() == numel):
                raise RuntimeError("fused_attention: %s must be a contiguous int32 device tensor of %d entries" % (name, numel))
    if presplit is not None and mode != "split":
        raise RuntimeError("fused_attention: pre-split Q / K planes belong to the split attention (mode %r)" % (mode,))
    if mode in ("split", "bf16"):
        L = capi.lib()
        ws_bytes = L.scp_vit_attention_split_workspace(b, n, heads)
        ip = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        if presplit is not None:
            # Q / K planes already in the workspace (written by the qkv GEMM's epilogue); the kernel re-lays V only
            if not (presplit.is_cuda and presplit.dtype == torch.uint8 and presplit.numel() >= ws_bytes):
                raise RuntimeError("fused_attention: presplit must be attention_workspace(b, n, heads, device)")
            capi.check(L.scp_vit_attention_split_forward_presplit(capi.dev_ptr(qkv, "qkv"), ip(out),
                                                                  ctypes.c_void_p(0 if out_planes is None else out_planes.blob.data_ptr()),
                                                                  b, n, heads, head_dim, float(scale), ip(q_rows), ip(q_count),
                                                                  ctypes.c_void_p(presplit.data_ptr()), presplit.numel(), capi.current_stream()),
                       "scp_vit_attention_split_forward_presplit")
            return out if out_planes is None else out_planes
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=qkv.device)
        capi.check(L.scp_vit_attention_split_forward(capi.dev_ptr(qkv, "qkv"), capi.dev_ptr(out, "out"), b, n, heads, head_dim,
                                                     float(scale), ip(q_rows), ip(q_count), int(mode == "split"),
                                                     ctypes.c_void_p(ws.data_ptr()), ws_bytes,
                                                     capi.current_stream()), "scp_vit_attention_split_forward")
        return out
    if q_rows is None:
        code = capi.lib().scp_vit_attention_forward(capi.dev_ptr(qkv, "qkv"), capi.dev_ptr(out, "out"), b, n, heads,
                                                    head_dim, float(scale), capi.current_stream())
    else:
        code = capi.lib().scp_vit_attention_forward_rows(capi.dev_ptr(qkv, "qkv"), capi.dev_ptr(out, "out"), b, n, heads, head_dim,
                                                         float(scale), ctypes.c_void_p(q_rows.data_ptr()),
                                                         ctypes.c_void_p(q_count.data_ptr()), capi.current_stream())
    capi.check(code, "scp_vit_attention_forward")
    return out


def fused_attention_bf16(qkv, b, n, heads, head_dim, scale):
    """configs[4] precision: bf16 qkv [b,n,3*heads*head_dim] (straight from the bf16 qkv GEMM) -> bf16 [b,n,heads*head_dim]
    on the bf16 matrix cores, fp32 softmax statistics (csrc/vit_attn_bf16.hip).  Forward only, GPU only."""
    import ctypes
    from . import capi
    if torch.is_grad_enabled() and qkv.requires_grad:
        raise RuntimeError("scp_amd.dino.fused_attention_bf16 is forward-only (frozen ViT)")
    if not qkv.is_cuda or qkv.dtype != torch.bfloat16:
        raise RuntimeError("fused_attention_bf16 expects a CUDA bfloat16 tensor")
    qkv = qkv.contiguous()
    out = torch.empty(b, n, heads * head_dim, dtype=torch.bfloat16, device=qkv.device)
    capi.check(capi.lib().scp_vit_attention_bf16_forward(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), b, n, heads,
                                                         head_dim, float(scale), capi.current_stream()), "scp_vit_attention_bf16_forward")
    return out


def add_layernorm(x, branch, norm):
    """(x + branch, LayerNorm(x + branch)) in one HBM pass (csrc/vit_norm.hip); branch may be None.
    x is updated in place (the residual stream of a frozen, no-grad forward).  GPU tensors only."""
    from . import capi
    if torch.is_grad_enabled() and (x.requires_grad or norm.weight.requires_grad):
        raise RuntimeError("scp_amd.dino.add_layernorm is forward-only (frozen ViT)")
    y = torch.empty_like(x)
    rows, c = x.numel() // x.shape[-1], x.shape[-1]
    code = capi.lib().scp_add_layernorm_forward(
        capi.dev_ptr(x, "x"), capi.opt_ptr(branch, "branch"), capi.dev_ptr(norm.weight, "gamma"),
        capi.dev_ptr(norm.bias, "beta"), float(norm.eps), rows, c,
        capi.dev_ptr(x, "sum_out") if branch is not None else capi.opt_ptr(None, "sum_out"),
        capi.dev_ptr(y, "y_out"), capi.current_stream())
    capi.check(code, "scp_add_layernorm_forward")
    return x, y


GEMM_BIAS, GEMM_BIAS_RESIDUAL, GEMM_LN, GEMM_LN_GELU = 0, 1, 2, 3      # include/scp_hip.h SCP_GEMM_*


GEMM_W_SPLIT3, GEMM_W_BF16 = 0x100, 0x200                                # include/scp_hip.h SCP_GEMM_W_SPLIT3 / _BF16
# "split": products on the bf16 matrix cores with exactly split operands (fp32 = h + m + l in bf16, six partial products, fp32
# accumulation -- csrc/gemm_core_split.h; as close to float64 as the fp32 matrix cores, ~1.5x their rate).
# "fp32": v_mfma_f32_32x32x2_f32.  SCP_VIT_GEMM=fp32 in the environment selects the latter for a whole process.
GEMM_MODE = os.environ.get("SCP_VIT_GEMM", "split")
# split mode: activations reach the next GEMM pre-split (bf16 planes written by the producing epilogue; csrc/vit_gemm.hip CORE 3).
# SCP_VIT_PRESPLIT=0: every GEMM splits its fp32 A operand in registers, as in round 3 (A/B switch).
PRESPLIT_ACTIVATIONS = os.environ.get("SCP_VIT_PRESPLIT", "1") == "1"
# the qkv projection's epilogue writes the split attention's Q / K operand planes (round 4; "0": the re-layout kernel makes all three)
QK_FROM_EPILOGUE = os.environ.get("SCP_VIT_QK_EPILOGUE", "1") == "1"


def split_weight(w):
    """planes [3,N,K] bf16 (h, m, l) with w = h + m + l exactly (scp_split_bf16x3).  The ViT blocks keep the planes of their
    frozen weights next to the LayerNorm-folded copies (_Block._folded); a call without `w_split` splits on the fly."""
    from . import capi
    w = w.detach().contiguous()
    planes = torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    capi.check(capi.lib().scp_split_bf16x3(capi.dev_ptr(w, "w"), ctypes.c_void_p(planes.data_ptr()), w.numel(),
                                           capi.current_stream()), "scp_split_bf16x3")
    return planes


class TiledPlanes:
    """a [rows, cols] fp32 operand as three bf16 planes in the TILED layout of csrc/gemm_core_split.h:
    [ceil(rows / 32)][cols / 16][3 planes][32 rows][16 k] -- every piece an LDS-DMA instruction moves is one contiguous KiB.
    `blob` is the flat bfloat16 storage; untile() gives the plane-major [3, rows, cols] view of it as a copy (tests)."""

    def __init__(self, rows, cols, device, blob=None):
        if cols % 16:
            raise RuntimeError("TiledPlanes: the column count must be a multiple of 16")
        self.rows, self.cols, self.rows_pad = rows, cols, (rows + 31) // 32 * 32
        self.blob = blob if blob is not None else torch.empty(3 * self.rows_pad * cols, dtype=torch.bfloat16, device=device)

    def zero_(self):
        self.blob.zero_()
        return self

    def untile(self):
        t = self.blob.view(self.rows_pad // 32, self.cols // 16, 3, 32, 16).permute(2, 0, 3, 1, 4)
        return t.reshape(3, self.rows_pad, self.cols)[:, :self.rows].contiguous()


def split_tiled(x):
    """x [rows, cols] fp32 -> TiledPlanes with x = h + m + l exactly (scp_split_bf16x3_tiled)"""
    from . import capi
    x = x.detach().contiguous()
    rows, cols = x.shape
    out = TiledPlanes(rows, cols, x.device)
    capi.check(capi.lib().scp_split_bf16x3_tiled(capi.dev_ptr(x, "x"), ctypes.c_void_p(out.blob.data_ptr()), rows, cols, capi.current_stream()),
               "scp_split_bf16x3_tiled")
    return out


def vit_linear(a, w, vec0, vec1=None, rowstat=None, resid=None, out=None, epilogue=GEMM_BIAS, rows=None, a_rows=None, c_rows=None,
               max_rows=None, mode=None, w_split=None, a_planes=None, out_planes=None, fp32_out=True, qk_planes=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T) on the matrix cores (csrc/vit_gemm.hip, include/scp_hip.h scp_vit_linear);
    `resid` may be `out` itself (in-place residual stream).  Forward only, GPU tensors only.  `mode` (default GEMM_MODE):
    "split" = bf16 matrix cores on exactly split operands (`w_split` = split_weight(w) if the caller keeps it), "fp32" = fp32
    matrix cores; both are fp32-accurate.
    Row selection made on the device (scp_vit_linear_rows): `rows` = int32 device scalar, only the first rows[0] GEMM rows
    are computed and the host never reads the count; GEMM row m reads a[a_rows[m]] and uses rowstat / resid / out row
    c_rows[m] (int32 index lists, None = identity); untouched rows of `out` keep their contents.
    Pre-split operands (split mode only, scp_vit_linear_planes): `a_planes` = TiledPlanes [rows, K], the A operand as the producing
    layer's epilogue left it (`a` may then be None; the main loop has no VALU split; `w_split` must then be the TiledPlanes of w);
    `out_planes` = TiledPlanes [rows, N] receives the result split the same way (the next layer's a_planes); `fp32_out=False` with
    out_planes: the fp32 result is not stored at all.
    `qk_planes` = (attention_workspace, tokens_per_image, heads, scale[, keep_fp32]) (split mode, the qkv projection with the
    GEMM_LN epilogue, scp_vit_linear_qkv): the Q and K thirds of the result are written as the split attention's operand planes into
    that workspace; keep_fp32 bit 0 / 1: the Q / K third is ALSO stored as fp32 (default 2: K only, which the attention's
    leftover-query kernel reads; the Q columns of `out` stay uninitialised)."""
    from . import capi
    if torch.is_grad_enabled() and ((a is not None and a.requires_grad) or w.requires_grad):
        raise RuntimeError("scp_amd.dino.vit_linear is forward-only (frozen ViT)")
    if a is None and a_planes is None:
        raise RuntimeError("vit_linear: no A operand")
    m, k = a.shape if a is not None else (a_planes.rows, a_planes.cols)
    n = w.shape[0]
    dev = a.device if a is not None else a_planes.blob.device
    if out is None and (fp32_out or out_planes is None):
        out = torch.empty(m, n, dtype=torch.float32, device=dev)
    L = capi.lib()
    mode = gemm_mode() if mode is None else mode
    if mode == "bf16":
        # configs[4] precision: W rounded to bf16 (one plane), A rounded in registers, one product, fp32 accumulate / output
        if w_split is None:
            w_split = w.detach().to(torch.bfloat16).contiguous()
        if not (w_split.is_cuda and w_split.dtype == torch.bfloat16 and w_split.is_contiguous() and tuple(w_split.shape) == (n, k)):
            raise RuntimeError("vit_linear: in bf16 mode w_split must be the contiguous [%d,%d] bfloat16 weight" % (n, k))
        w_ptr = ctypes.c_void_p(w_split.data_ptr())
        epilogue = epilogue | GEMM_W_BF16
    elif mode == "split":
        if a_planes is not None:
            if w_split is None:
                w_split = split_tiled(w)
            w_ptr = None                                  # set with the other plane operands below
        else:
            if w_split is None:
                w_split = split_weight(w)
            if not (torch.is_tensor(w_split) and w_split.is_cuda and w_split.dtype == torch.bfloat16 and w_split.is_contiguous()
                    and tuple(w_split.shape) == (3, n, k)):
                raise RuntimeError("vit_linear: w_split must be the contiguous [3,%d,%d] bfloat16 planes of w" % (n, k))
            w_ptr = ctypes.c_void_p(w_split.data_ptr())
        epilogue = epilogue | GEMM_W_SPLIT3
    elif mode == "fp32":
        w_ptr = capi.dev_ptr(w, "w")
    else:
        raise RuntimeError("vit_linear: unknown mode %r" % (mode,))
    if a_planes is not None or out_planes is not None or qk_planes is not None:
        if mode != "split":
            raise RuntimeError("vit_linear: operand planes need the split main loop (mode %r)" % (mode,))
        for t, name, cols in ((a_planes, "a_planes", k), (out_planes, "out_planes", n)):
            if t is not None and not (isinstance(t, TiledPlanes) and t.cols == cols and t.blob.is_cuda):
                raise RuntimeError("vit_linear: %s must be TiledPlanes with %d columns" % (name, cols))
        if a_planes is not None:
            if not (isinstance(w_split, TiledPlanes) and (w_split.rows, w_split.cols) == (n, k)):
                raise RuntimeError("vit_linear: with a_planes, w_split must be split_tiled(w)")
            w_ptr = ctypes.c_void_p(w_split.blob.data_ptr())
        for t, name in ((rows, "rows"), (a_rows, "a_rows"), (c_rows, "c_rows")):
            if t is not None and not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
                raise RuntimeError("vit_linear: %s must be a contiguous int32 device tensor" % name)
        ip = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        bp = lambda t: ctypes.c_void_p(0 if t is None else t.blob.data_ptr())
        if rows is not None and max_rows is None:
            max_rows = m
        if qk_planes is not None:
            ws, tokens, heads, scale = qk_planes[:4]
            if rows is not None or a_rows is not None or c_rows is not None or out_planes is not None or resid is not None:
                raise RuntimeError("vit_linear: qk_planes goes with the plain qkv projection (no row selection, no output planes)")
            code = L.scp_vit_linear_qkv(ip(a if a_planes is None else None), bp(a_planes), 0 if a_planes is None else a_planes.rows_pad, w_ptr,
                                        capi.dev_ptr(vec0, "vec0"), capi.opt_ptr(vec1, "vec1"), capi.opt_ptr(rowstat, "rowstat"), ip(out),
                                        m, n, k, epilogue, ctypes.c_void_p(ws.data_ptr()), int(tokens), int(heads), float(scale),
                                        int(qk_planes[4]) if len(qk_planes) > 4 else 2, capi.current_stream())
            capi.check(code, "scp_vit_linear_qkv")
            return out
        code = L.scp_vit_linear_planes(ip(a if a_planes is None else None), bp(a_planes), 0 if a_planes is None else a_planes.rows_pad, w_ptr,
                                       capi.dev_ptr(vec0, "vec0"), capi.opt_ptr(vec1, "vec1"), capi.opt_ptr(rowstat, "rowstat"),
                                       capi.opt_ptr(resid, "resid"), ip(out), bp(out_planes),
                                       0 if out_planes is None else out_planes.rows_pad, ip(rows), m if rows is None else max_rows,
                                       ip(a_rows), ip(c_rows), n, k, epilogue, capi.current_stream())
        capi.check(code, "scp_vit_linear_planes")
        return out
    if rows is None:
        code = L.scp_vit_linear(capi.dev_ptr(a, "a"), w_ptr, capi.dev_ptr(vec0, "vec0"),
                                capi.opt_ptr(vec1, "vec1"), capi.opt_ptr(rowstat, "rowstat"),
                                capi.opt_ptr(resid, "resid"), capi.dev_ptr(out, "out"), m, n, k, epilogue,
                                capi.current_stream())
    else:
        for t, name in ((rows, "rows"), (a_rows, "a_rows"), (c_rows, "c_rows")):
            if t is not None and not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
                raise RuntimeError("vit_linear: %s must be a contiguous int32 device tensor" % name)
        max_rows = min(m, out.shape[0]) if max_rows is None else max_rows
        ip = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
        code = L.scp_vit_linear_rows(capi.dev_ptr(a, "a"), w_ptr, capi.dev_ptr(vec0, "vec0"),
                                     capi.opt_ptr(vec1, "vec1"), capi.opt_ptr(rowstat, "rowstat"),
                                     capi.opt_ptr(resid, "resid"), capi.dev_ptr(out, "out"), ip(rows), max_rows, ip(a_rows),
                                     ip(c_rows), n, k, epilogue, capi.current_stream())
    capi.check(code, "scp_vit_linear")
    return out


def row_mean_rstd(x, eps):
    """[M,2] (mean, 1/sqrt(biased var + eps)) per row of x[M,C] -- nn.LayerNorm's statistics (csrc/vit_gemm.hip)"""
    from . import capi
    m, c = x.shape
    st = torch.empty(m, 2, dtype=torch.float32, device=x.device)
    capi.check(capi.lib().scp_row_mean_rstd(capi.dev_ptr(x, "x"), capi.dev_ptr(st, "stats"), m, c, float(eps),
                                            capi.current_stream()), "scp_row_mean_rstd")
    return st


def fold_layernorm(norm, weight, bias):
    """LayerNorm(x) W^T + b = rstd * (x (gamma o W)^T - mean * s) + t  with s = rowsum(gamma o W), t = W beta + b.
    The ViT is frozen, so this is computed once per weight version (float64 sums, stored fp32)."""
    w64 = weight.double()
    wg = (w64 * norm.weight.double()[None]).float().contiguous()
    s = wg.double().sum(1).float().contiguous()
    t = (w64 @ norm.bias.double() + bias.double()).float().contiguous()
    return wg, s, t


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, pending=None):
        """x: residual stream, pending: output of the previous block's MLP not yet added to it.
        Returns (x, pending') -- every residual add is fused into the LayerNorm that follows it."""
        x, y = add_layernorm(x, pending, self.norm1)
        x, y = add_layernorm(x, self.attn(y), self.norm2)
        return x, self.mlp(y)

    # ---- fused path (fp32, GPU): four GEMM launches + two row-statistics launches + attention per block -------------
    def _folded(self):
        """(gamma1 o Wqkv, s, t), (gamma2 o W1, s, t), rebuilt when any of the frozen tensors changes (load_state_dict);
        the bf16 planes of the five weight matrices the fused path multiplies with are rebuilt with them (self._planes)"""
        src = (self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias,
               self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.attn.proj.weight, self.mlp.fc2.weight)
        key = tuple((t.data_ptr(), t._version, str(t.device)) for t in src) + (gemm_mode(),)
        if getattr(self, "_fold_key", None) != key:
            with torch.no_grad():
                self._fold = (fold_layernorm(self.norm1, self.attn.qkv.weight, self.attn.qkv.bias),
                              fold_layernorm(self.norm2, self.mlp.fc1.weight, self.mlp.fc1.bias))
                c = self.norm1.weight.shape[0]
                wq, w1 = self._fold[0][0], self._fold[1][0]
                if gemm_mode() == "split" and wq.is_cuda:
                    self._planes = dict(qkv=split_weight(wq), k=split_weight(wq[c:2 * c]), fc1=split_weight(w1),
                                        proj=split_weight(self.attn.proj.weight), fc2=split_weight(self.mlp.fc2.weight))
                    if PRESPLIT_ACTIVATIONS:     # tiled planes: the W operand of the layers whose A operand arrives pre-split
                        self._planes.update(qkv_t=split_tiled(wq), k_t=split_tiled(wq[c:2 * c].contiguous()), fc1_t=split_tiled(w1),
                                            fc2_t=split_tiled(self.mlp.fc2.weight), proj_t=split_tiled(self.attn.proj.weight))
                elif gemm_mode() == "bf16" and wq.is_cuda:
                    rnd = lambda t: t.detach().to(torch.bfloat16).contiguous()
                    self._planes = dict(qkv=rnd(wq), k=rnd(wq[c:2 * c]), fc1=rnd(w1), proj=rnd(self.attn.proj.weight),
                                        fc2=rnd(self.mlp.fc2.weight))
                else:
                    self._planes = dict(qkv=None, k=None, fc1=None, proj=None, fc2=None)
            self._fold_key = key
        return self._fold

    def forward_fused(self, x2d, b, n, x3=None):
        """x2d [b*n, dim] residual stream, updated IN PLACE:  x += proj(attn(LN1 x)); x += fc2(gelu(fc1(LN2 x)))
        (vision_transformer_flexible.py:126-132).
        x3 (split mode): the bf16 planes [3, b*n, dim] of x2d (x = h + m + l exactly), kept in step with it: every GEMM that
        consumes the residual stream or the MLP's hidden activation reads its A operand pre-split (no VALU split in its main loop),
        and the epilogues that produce them -- proj + residual, fc1 + GELU, fc2 + residual -- write the planes once (the hidden
        activation exists ONLY as planes)."""
        (wq, sq, tq), (w1, s1, t1) = self._folded()
        a = self.attn
        sp = self._planes
        pl = x3 is not None and gemm_mode() == "split"
        # the qkv projection's epilogue writes the attention's Q / K operand planes itself (the re-layout pass then moves V only)
        ws = attention_workspace(b, n, a.num_heads, x2d.device) if (pl and attn_mode() == "split" and QK_FROM_EPILOGUE) else None
        qkv = vit_linear(x2d, wq, sq, tq, row_mean_rstd(x2d, self.norm1.eps), epilogue=GEMM_LN, w_split=sp["qkv_t" if pl else "qkv"],
                         a_planes=x3 if pl else None, qk_planes=None if ws is None else (ws, n, a.num_heads, a.scale))
        if ws is not None:
            # the attention writes its result as the proj GEMM's pre-split A operand (no fp32 copy at all)
            y3 = fused_attention(qkv.view(b, n, -1), b, n, a.num_heads, x2d.shape[1] // a.num_heads, a.scale, presplit=ws,
                                 out_planes=TiledPlanes(b * n, x2d.shape[1], x2d.device))
            vit_linear(None, a.proj.weight, a.proj.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["proj_t"], a_planes=y3,
                       out_planes=x3)
        else:
            y = fused_attention(qkv.view(b, n, -1), b, n, a.num_heads, x2d.shape[1] // a.num_heads, a.scale)
            vit_linear(y.view(b * n, -1), a.proj.weight, a.proj.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["proj"],
                       out_planes=x3 if pl else None)
        if pl:
            h3 = TiledPlanes(b * n, w1.shape[0], x2d.device)
            vit_linear(None, w1, s1, t1, row_mean_rstd(x2d, self.norm2.eps), epilogue=GEMM_LN_GELU, w_split=sp["fc1_t"], a_planes=x3,
                       out_planes=h3, fp32_out=False)
            vit_linear(None, self.mlp.fc2.weight, self.mlp.fc2.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["fc2_t"],
                       a_planes=h3, out_planes=x3)
            return x2d
        h = vit_linear(x2d, w1, s1, t1, row_mean_rstd(x2d, self.norm2.eps), epilogue=GEMM_LN_GELU, w_split=sp["fc1"])
        vit_linear(h, self.mlp.fc2.weight, self.mlp.fc2.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["fc2"])
        return x2d

    def tail_keys_fused(self, x2d, b, n, keep, key_block, x3=None):
        """The last block before the key layer, and the keys themselves, for the tokens in `keep` only (bool [b, n]):
        after this block's attention nothing mixes tokens any more, so proj / MLP of this block and LN1 + K of `key_block`
        are needed only for the tokens whose keys are consumed (pretrained_corr.py:85-89 masks every other token out of the
        matching).  The QKV projection still runs on all tokens (keys / values of every token feed the kept queries); the
        attention runs for the kept queries only.  The GEMMs take their row count and row index list from the device, and the
        keys of all other tokens are returned as zeros.  Same values for the kept tokens as the full path (row-wise identical GEMMs).
        Returns the keys token-major: [b, n, heads * d]; with `x3` (the residual stream's bf16 planes, see forward_fused) the keys'
        own planes ride along as `._scp_planes` -- the pre-split W operand of the mutual-nearest-neighbour kernel."""
        (wq, sq, tq), (w1, s1, t1) = self._folded()
        a = self.attn
        c = x2d.shape[1]
        sp = self._planes
        pl = x3 is not None and gemm_mode() == "split"
        ws = attention_workspace(b, n, a.num_heads, x2d.device) if (pl and attn_mode() == "split" and QK_FROM_EPILOGUE) else None
        qkv = vit_linear(x2d, wq, sq, tq, row_mean_rstd(x2d, self.norm1.eps), epilogue=GEMM_LN, w_split=sp["qkv_t" if pl else "qkv"],
                         a_planes=x3 if pl else None, qk_planes=None if ws is None else (ws, n, a.num_heads, a.scale))
        # attention: keys / values of all tokens, queries only for the kept ones (per image, compacted to the front of the
        # query slots; their outputs land on their own rows)
        k8 = keep.to(torch.uint8)
        q_rows = torch.argsort(k8, dim=1, descending=True, stable=True).to(torch.int32)
        q_count = keep.sum(1, dtype=torch.int32)
        y3 = None
        if ws is not None:
            y3 = fused_attention(qkv.view(b, n, -1), b, n, a.num_heads, c // a.num_heads, a.scale, q_rows, q_count, presplit=ws,
                                 out_planes=TiledPlanes(b * n, c, x2d.device))
        else:
            y = fused_attention(qkv.view(b, n, -1), b, n, a.num_heads, c // a.num_heads, a.scale, q_rows, q_count).view(b * n, c)
        flat = keep.reshape(-1)
        idx = torch.argsort(flat.to(torch.uint8), descending=True, stable=True).to(torch.int32)   # kept rows first, original order
        rows = flat.sum(dtype=torch.int32).reshape(1)
        m = b * n
        sel = dict(rows=rows, max_rows=m, a_rows=idx, c_rows=idx)
        # x[idx] += proj(y[idx]);  h = gelu(fc1(LN2 x[idx]));  x[idx] += fc2(h);  k[idx] = Wk LN1(x[idx]) -- rows addressed
        # through the index list inside the GEMM (no gather / scatter copies); statistics are taken for all rows (11 us)
        if y3 is not None:
            vit_linear(None, a.proj.weight, a.proj.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["proj_t"], a_planes=y3,
                       out_planes=x3, **sel)
        else:
            vit_linear(y, a.proj.weight, a.proj.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["proj"],
                       out_planes=x3 if pl else None, **sel)
        (kq, ks, kt), _ = key_block._folded()
        k = torch.zeros(m, c, dtype=torch.float32, device=x2d.device)
        if pl:
            h3 = TiledPlanes(m, w1.shape[0], x2d.device)                                       # only the kept rows are written / read
            vit_linear(None, w1, s1, t1, row_mean_rstd(x2d, self.norm2.eps), epilogue=GEMM_LN_GELU, w_split=sp["fc1_t"], a_planes=x3,
                       out_planes=h3, fp32_out=False, **sel)
            vit_linear(None, self.mlp.fc2.weight, self.mlp.fc2.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["fc2_t"],
                       a_planes=h3, out_planes=x3, **sel)
            k3 = TiledPlanes(m, c, x2d.device).zero_()
            vit_linear(None, kq[c:2 * c], ks[c:2 * c], kt[c:2 * c], row_mean_rstd(x2d, key_block.norm1.eps), out=k, epilogue=GEMM_LN,
                       w_split=key_block._planes["k_t"], a_planes=x3, out_planes=k3, **sel)
            k = k.view(b, n, c)
            k._scp_planes = k3
            return k
        h = torch.empty(m, w1.shape[0], dtype=torch.float32, device=x2d.device)           # only the kept rows are written / read
        vit_linear(x2d, w1, s1, t1, row_mean_rstd(x2d, self.norm2.eps), out=h, epilogue=GEMM_LN_GELU, w_split=sp["fc1"], **sel)
        vit_linear(h, self.mlp.fc2.weight, self.mlp.fc2.bias, resid=x2d, out=x2d, epilogue=GEMM_BIAS_RESIDUAL, w_split=sp["fc2"], **sel)
        vit_linear(x2d, kq[c:2 * c], ks[c:2 * c], kt[c:2 * c], row_mean_rstd(x2d, key_block.norm1.eps), out=k, epilogue=GEMM_LN,
                   w_split=key_block._planes["k"], **sel)
        return k.view(b, n, c)

    def keys_fused(self, x2d, b, n, x3=None):
        """K third of qkv(LN1(x)), token-major [b, n, heads * d]  (the only part of block 9 the DINO features need)"""
        (wq, sq, tq), _ = self._folded()
        c = x2d.shape[1]
        pl = x3 is not None and gemm_mode() == "split"
        k3 = TiledPlanes(b * n, c, x2d.device) if pl else None
        k = vit_linear(None if pl else x2d, wq[c:2 * c], sq[c:2 * c], tq[c:2 * c], row_mean_rstd(x2d, self.norm1.eps), epilogue=GEMM_LN,
                       w_split=self._planes["k_t" if pl else "k"], a_planes=x3 if pl else None, out_planes=k3)
        k = k.view(b, n, c)
        if pl:
            k._scp_planes = k3
        return k


class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, embed_dim, img_size=224):
        super().__init__()
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        if self._own_gemm_ok(x):
            return self._forward_gemm(x)
        return self.proj(x).flatten(2).transpose(1, 2)

    def _own_gemm_ok(self, x):
        """the frozen ViT on the GPU: the patch embedding is one [B * patches, 3 p^2] x [3 p^2, embed] product on the build's own GEMM
        (vision_transformer_flexible.py:134-149 is a p x p / stride-p convolution = exactly that), no library kernel"""
        w = self.proj.weight
        k = w.shape[1] * w.shape[2] * w.shape[3]
        return (x.is_cuda and x.dim() == 4 and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))
                and w.dtype == torch.float32 and k % 16 == 0 and x.shape[2] % self.patch_size == 0 and x.shape[3] % self.patch_size == 0
                and (MIXED_OWN_KERNELS or not MIXED_BF16) and os.environ.get("SCP_PATCH_EMBED", "own") == "own")

    def _forward_gemm(self, x):
        b, c, h, w = x.shape
        p, n = self.patch_size, self.proj.weight.shape[0]
        # im2col of non-overlapping patches is a pure permutation: rows = (image, patch row, patch column), K = (channel, ky, kx), the
        # order of proj.weight.reshape(embed, -1); the result IS the token-major [B, patches, embed] the blocks consume
        a = x.float().reshape(b, c, h // p, p, w // p, p).permute(0, 2, 4, 1, 3, 5).reshape(b * (h // p) * (w // p), c * p * p)
        wt = self.proj.weight
        mode = "fp32" if GEMM_MODE == "fp32" else "split"       # fp32-accurate in every precision mode (3 % of the ViT's flops)
        key = (wt.data_ptr(), wt._version, mode)
        cache = self.__dict__.get("_w2d")
        if cache is None or cache[0] != key:
            w2d = wt.detach().reshape(n, c * p * p).contiguous()
            cache = self.__dict__["_w2d"] = (key, w2d, split_weight(w2d) if mode == "split" else None)
        out = vit_linear(a, cache[1], self.proj.bias.detach(), epilogue=GEMM_BIAS, w_split=cache[2], mode=mode)
        return out.view(b, -1, n)


class VisionTransformer(nn.Module):
    def __init__(self, patch_size=8, embed_dim=384, depth=12, num_heads=6):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)
        self._pos_cache = {}

    def interpolate_pos_encoding(self, npatch, w, h):
        n = self.pos_embed.shape[1] - 1
        if npatch == n and w == h:
            return self.pos_embed
        key = (w, h, self.pos_embed.device, self.pos_embed._version)
        if key not in self._pos_cache:
            dim = self.pos_embed.shape[-1]
            side = int(math.sqrt(n))
            w0, h0 = w // self.patch_embed.patch_size + 0.1, h // self.patch_embed.patch_size + 0.1
            grid = F.interpolate(self.pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2),
                                 scale_factor=(w0 / math.sqrt(n), h0 / math.sqrt(n)), mode="bicubic")
            assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
            grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
            self._pos_cache = {key: torch.cat((self.pos_embed[:, :1], grid), 1).detach()}
        return self._pos_cache[key]

    def prepare_tokens(self, x):
        b, _, w, h = x.shape
        tok = self.patch_embed(x)
        tok = torch.cat((self.cls_token.expand(b, -1, -1), tok), 1)
        return tok + self.interpolate_pos_encoding(tok.shape[1] - 1, w, h)

    def key_features(self, x, layer=9, keep=None):
        """keys of block `layer`: [b, heads, tokens, d] (a view of key_tokens)"""
        k = self.key_tokens(x, layer, keep)
        b, n, c = k.shape
        heads = self.blocks[layer].attn.num_heads
        return k.view(b, n, heads, c // heads).permute(0, 2, 1, 3)

    def key_tokens(self, x, layer=9, keep=None):
        """keys of block `layer`, token-major as the K projection leaves them: [b, tokens, heads * d] contiguous, channel =
        head * 64 + d (dino.py:102-108).  `keep` (bool [b, tokens - 1], patch tokens): only these tokens' keys are needed -- the
        others come back as zeros (fused GPU path; ignored elsewhere, where all keys are computed)"""
        if x.is_cuda and (MIXED_OWN_KERNELS or not MIXED_BF16):   # fused path; under MIXED_BF16 the same kernels, operands rounded to bf16
            tok = self.prepare_tokens(x).float().contiguous()
            b, n, c = tok.shape
            x2d = tok.view(b * n, c)
            # split main loop: the residual stream also lives as bf16 planes, the pre-split A operand of the layers that read it
            x3 = split_tiled(x2d) if (gemm_mode() == "split" and PRESPLIT_ACTIVATIONS) else None
            if keep is not None and layer >= 1:
                for blk in self.blocks[:layer - 1]:
                    blk.forward_fused(x2d, b, n, x3)
                keep_tok = torch.cat((torch.zeros(b, 1, dtype=torch.bool, device=x.device), keep.reshape(b, n - 1).bool()), 1)
                return self.blocks[layer - 1].tail_keys_fused(x2d, b, n, keep_tok, self.blocks[layer], x3)
            for blk in self.blocks[:layer]:
                blk.forward_fused(x2d, b, n, x3)
            return self.blocks[layer].keys_fused(x2d, b, n, x3)
        tok, pending = self.prepare_tokens(x).contiguous(), None
        for blk in self.blocks[:layer]:
            tok, pending = blk(tok, pending)
        blk = self.blocks[layer]
        tok, y = add_layernorm(tok, pending, blk.norm1)
        k = blk.attn.keys(y)                                   # b,h,n,d
        return k.permute(0, 2, 1, 3).reshape(k.shape[0], k.shape[2], -1)

    @torch.no_grad()
    def forward(self, x):
        """class-token embedding of the full 12-block ViT (not used by the training step)"""
        tok, pending = self.prepare_tokens(x).contiguous(), None
        for blk in self.blocks:
            tok, pending = blk(tok, pending)
        return add_layernorm(tok, pending, self.norm)[1][:, 0]


def vit_small(patch_size=8):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6)


class DINO(nn.Module):
    def __init__(self, pretrain_path="pretrain/dino_deitsmall8_pretrain.pth"):
        super().__init__()
        self.patch_size, self.feat_layer, self.num_patches = 8, 9, 32
        self.model_name, self.pretrain_path = "vit_small", pretrain_path
        self.model = vit_small(self.patch_size)
        if os.path.exists(pretrain_path):
            self.model.load_state_dict(torch.load(pretrain_path, map_location="cpu"))
        elif not ALLOW_RANDOM_INIT:
            raise FileNotFoundError("%s not found (set scp_amd.dino.ALLOW_RANDOM_INIT for synthetic weights)" % pretrain_path)
        self.model.eval()

    def train(self, mode=True):
        return super().train(False)  # frozen feature extractor, always eval (pretrained_corr.py:21)

    @torch.no_grad()
    def key_tokens(self, img, keep=None):
        """the block-9 keys token-major, class token included at row 0: [b, 1 + side*side, 384] contiguous.  This is the layout the
        mutual-nearest-neighbour matching consumes (scp_mutual_nn_fused: a token's key is one K-contiguous GEMM row); forward()
        is its channel-major view for callers that want the reference's [b, 384, side, side] map."""
        return self.model.key_tokens(img, self.feat_layer, keep)

    @torch.no_grad()
    def forward(self, img, keep=None):
        """`keep` (bool [b, side*side] or [b, side, side]): the patch tokens whose features will be read; the others may come
        back as zeros (VisionTransformer.key_tokens)"""
        k = self.key_tokens(img, keep)[:, 1:, :]                                 # drop cls: b,t,c
        b, t, c = k.shape
        side = int(math.sqrt(t))
        return k.transpose(1, 2).reshape(b, c, side, side)
