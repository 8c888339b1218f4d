"""oracle/vit.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement of the ViT pieces the HIP kernels replace, written the way the reference writes them
(third-party/zsp/zsp/method/vision_transformer_flexible.py): Attention.forward :85-101 with the score
matrix materialised.  Pinned end-to-end by the DINO key-feature fixture recorded from the reference
(tests/golden/step_*.npz: dino_feat_sub / dino_feat_stats)."""
import torch


def attention_oracle(qkv_linear_out, num_heads, scale):
    """qkv_linear_out [B,N,3*C] (output of self.qkv) -> [B,N,C]"""
    B, N, C3 = qkv_linear_out.shape
    C = C3 // 3
    qkv = qkv_linear_out.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B, N, C)


def add_layernorm_oracle(x, branch, weight, bias, eps):
    """x = x + drop_path(branch); y = norm(x)  (vision_transformer_flexible.py:117-120)"""
    s = x if branch is None else x + branch
    return s, torch.nn.functional.layer_norm(s, (s.shape[-1],), weight, bias, eps)


def block_oracle(x, p, num_heads, eps=1e-6):
    """Block.forward (vision_transformer_flexible.py:126-132) with Attention :85-101 and Mlp :54-70 spelled out:
    x [B,N,C]; p = dict(norm1_w, norm1_b, qkv_w, qkv_b, proj_w, proj_b, norm2_w, norm2_b, fc1_w, fc1_b, fc2_w, fc2_b).
    Evaluated in the dtype of x (float64 for a reference value)."""
    F = torch.nn.functional
    C = x.shape[-1]
    y = F.layer_norm(x, (C,), p["norm1_w"], p["norm1_b"], eps)
    y = attention_oracle(F.linear(y, p["qkv_w"], p["qkv_b"]), num_heads, (C // num_heads) ** -0.5)
    x = x + F.linear(y, p["proj_w"], p["proj_b"])
    y = F.layer_norm(x, (C,), p["norm2_w"], p["norm2_b"], eps)
    return x + F.linear(F.gelu(F.linear(y, p["fc1_w"], p["fc1_b"])), p["fc2_w"], p["fc2_b"])


def block_keys_oracle(x, p, num_heads, eps=1e-6):
    """k of Block.return_qkv (:121-124): the K third of qkv(norm1(x)) as [B, heads, N, d]"""
    F = torch.nn.functional
    B, N, C = x.shape
    qkv = F.linear(F.layer_norm(x, (C,), p["norm1_w"], p["norm1_b"], eps), p["qkv_w"], p["qkv_b"])
    return qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)[1]
