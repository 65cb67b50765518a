"""Oracle (test infrastructure only): BLIP image encoder (timm-style ViT) + vision_proj.

PINNED at block and tower level: tests/golden/make_golden_blipvit.py imports /root/reference/blip4cir/vit.py with
import-only stubs for its timm / fairscale imports and runs the reference's own Block / Attention / Mlp (vit.py:23-112)
and VisionTransformer.forward (vit.py:183-197: cls concat, pos_embed, blocks, final norm over all tokens);
tests/test_oracle_golden.py checks `block()` and `img_embed()` below against those vectors (tests/golden/blip_vit.npz).
NOT pinned (by reading): timm's PatchEmbed (a stride = kernel Conv2d WITH bias -> flatten(2).transpose(1, 2); timm is
not part of /root/reference) and blip_cir.py:62 (normalize(vision_proj(x[:, 0])); blip_cir.py needs tokenizer files).

Follows: vit.py:91-112 (pre-LN block, LayerNorm eps 1e-6 via partial(nn.LayerNorm, eps=1e-6) at :143), :46-88 (fused qkv
Linear with bias, rows ordered q,k,v by the reshape at :73, scale head_dim^-0.5), :23-43 (fc1, exact GELU, fc2)."""
import math

import torch
import torch.nn.functional as F


def attention(sd, b, h, heads):
    """vit.py:71-88 (Attention.forward) on the normalised input h [B, N, W]; b = key prefix of the block."""
    B, _, W = h.shape
    hd = W // heads
    qkv = (h @ sd[b + "attn.qkv.weight"].t() + sd[b + "attn.qkv.bias"]).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
    a = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * hd ** -0.5, dim=-1) @ qkv[2]
    return a.transpose(1, 2).reshape(B, -1, W) @ sd[b + "attn.proj.weight"].t() + sd[b + "attn.proj.bias"]


def block(sd, b, x, heads):
    """vit.py:108-112 (Block.forward, drop_path = Identity at inference)."""
    W = x.shape[-1]
    x = x + attention(sd, b, F.layer_norm(x, (W,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], 1e-6), heads)
    h = F.layer_norm(x, (W,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], 1e-6)
    u = F.gelu(h @ sd[b + "mlp.fc1.weight"].t() + sd[b + "mlp.fc1.bias"])
    return x + u @ sd[b + "mlp.fc2.weight"].t() + sd[b + "mlp.fc2.bias"]


def img_embed(sd, image, heads, prefix="visual_encoder."):
    """-> (tokens [B,S,W] after the final norm, pooled = normalize(vision_proj(tokens[:,0])))"""
    w = sd[prefix + "patch_embed.proj.weight"]
    W, p = w.shape[0], w.shape[-1]
    x = F.conv2d(image.float(), w, sd[prefix + "patch_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    B = x.shape[0]
    x = torch.cat([sd[prefix + "cls_token"].expand(B, -1, -1), x], dim=1) + sd[prefix + "pos_embed"][:, :x.shape[1] + 1]
    layers = len({k[len(prefix):].split(".")[1] for k in sd if k.startswith(prefix + "blocks.")})
    for l in range(layers):
        x = block(sd, f"{prefix}blocks.{l}.", x, heads)
    x = F.layer_norm(x, (W,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6)
    pooled = F.normalize(x[:, 0] @ sd["vision_proj.weight"].t() + sd["vision_proj.bias"], dim=-1)
    return x, pooled


def synthetic_state_dict(W=128, layers=2, patch=16, res=64, proj=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, std=0.05: torch.randn(*s, generator=g) * std
    S = (res // patch) ** 2 + 1
    sd = {"visual_encoder.patch_embed.proj.weight": r(W, 3, patch, patch), "visual_encoder.patch_embed.proj.bias": r(W),
          "visual_encoder.cls_token": r(1, 1, W), "visual_encoder.pos_embed": r(1, S, W),
          "visual_encoder.norm.weight": 1 + r(W), "visual_encoder.norm.bias": r(W),
          "vision_proj.weight": r(proj, W, std=0.1), "vision_proj.bias": r(proj)}
    for l in range(layers):
        b = f"visual_encoder.blocks.{l}."
        sd.update({b + "norm1.weight": 1 + r(W), b + "norm1.bias": r(W), b + "attn.qkv.weight": r(3 * W, W, std=0.1),
                   b + "attn.qkv.bias": r(3 * W), b + "attn.proj.weight": r(W, W, std=0.1), b + "attn.proj.bias": r(W),
                   b + "norm2.weight": 1 + r(W), b + "norm2.bias": r(W), b + "mlp.fc1.weight": r(4 * W, W, std=0.1),
                   b + "mlp.fc1.bias": r(4 * W), b + "mlp.fc2.weight": r(W, 4 * W, std=0.1), b + "mlp.fc2.bias": r(W)})
    return sd
