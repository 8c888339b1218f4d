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
