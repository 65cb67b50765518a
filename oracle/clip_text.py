"""Oracle: CLIP text tower forward (CPU, fp32, torch autograd gives the backward).

Restates clip4cir/clip/model.py:345-358 (CLIP.encode_text), :171-192
(ResidualAttentionBlock), :157-163 (LayerNorm with fp32 up-cast), :166-168 (QuickGELU) and
:330-336 (causal additive -inf mask).  torch.nn.MultiheadAttention (torch, third party) is
restated from its published definition: packed in-projection with rows ordered q,k,v,
per-head softmax(q k^T / sqrt(head_dim) + mask) v, then out-projection.

Parameters are passed as a plain dict keyed like the CLIP state-dict (no ``clip.`` prefix).
"""
import math
import torch
import torch.nn.functional as F


def text_cfg_from_state_dict(sd):
    """Shape inference, restating clip4cir/clip/model.py:420-426."""
    width = sd["ln_final.weight"].shape[0]
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    return {
        "embed_dim": sd["text_projection"].shape[1],
        "context_length": sd["positional_embedding"].shape[0],
        "vocab_size": sd["token_embedding.weight"].shape[0],
        "width": width,
        "heads": width // 64,
        "layers": layers,
    }


def quick_gelu(x):
    # clip4cir/clip/model.py:166-168
    return x * torch.sigmoid(1.702 * x)


def attention(x, w_in, b_in, w_out, b_out, heads, mask):
    """x: [B, L, W].  Self attention as nn.MultiheadAttention(W, heads) computes it
    (clip4cir/clip/model.py:175,186-187)."""
    B, L, W = x.shape
    hd = W // heads
    qkv = x @ w_in.t() + b_in                      # [B, L, 3W], rows of w_in ordered q, k, v
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(B, L, heads, hd).transpose(1, 2)    # [B, H, L, hd]
    k = k.view(B, L, heads, hd).transpose(1, 2)
    v = v.view(B, L, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, L, W)
    return o @ w_out.t() + b_out


def residual_block(x, sd, prefix, heads, mask, eps=1e-5, act=quick_gelu):
    # clip4cir/clip/model.py:189-192: pre-LN block
    W = x.shape[-1]
    h = F.layer_norm(x, (W,), sd[prefix + "ln_1.weight"], sd[prefix + "ln_1.bias"], eps)
    x = x + attention(h, sd[prefix + "attn.in_proj_weight"], sd[prefix + "attn.in_proj_bias"],
                      sd[prefix + "attn.out_proj.weight"], sd[prefix + "attn.out_proj.bias"],
                      heads, mask)
    h = F.layer_norm(x, (W,), sd[prefix + "ln_2.weight"], sd[prefix + "ln_2.bias"], eps)
    u = act(h @ sd[prefix + "mlp.c_fc.weight"].t() + sd[prefix + "mlp.c_fc.bias"])
    x = x + u @ sd[prefix + "mlp.c_proj.weight"].t() + sd[prefix + "mlp.c_proj.bias"]
    return x


def causal_mask(L, dtype=torch.float32):
    # clip4cir/clip/model.py:330-336: -inf strictly above the diagonal
    return torch.full((L, L), float("-inf"), dtype=dtype).triu_(1)


def encode_text(sd, ids, return_hidden=False):
    """ids: int tensor [B, L] -> un-normalised text features [B, D].

    clip4cir/clip/model.py:345-358.  The EOT position is ``ids.argmax(-1)`` (the EOT token
    has the highest id in the vocabulary)."""
    cfg = text_cfg_from_state_dict(sd)
    ids = ids.long()
    B, L = ids.shape
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][:L]
    mask = causal_mask(L, x.dtype)
    hidden = [x]
    for i in range(cfg["layers"]):
        x = residual_block(x, sd, f"transformer.resblocks.{i}.", cfg["heads"], mask)
        hidden.append(x)
    x = F.layer_norm(x, (cfg["width"],), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    feats = x[torch.arange(B), ids.argmax(dim=-1)] @ sd["text_projection"]
    if return_hidden:
        return feats, hidden
    return feats


def synthetic_text_state_dict(width, layers, embed_dim, vocab=49408, ctx=77, seed=0, device="cpu"):
    """Seeded random text-tower weights with the distributions of
    CLIP.initialize_parameters (clip4cir/clip/model.py:301-328): used by bench.py and the
    parity tests.  Generated on the CPU generator so the
    values are bit-identical on every machine."""
    g = torch.Generator(device="cpu").manual_seed(seed)

    def normal(shape, std):
        return (torch.randn(shape, generator=g) * std).to(device)

    sd = {}
    sd["token_embedding.weight"] = normal((vocab, width), 0.02)
    sd["positional_embedding"] = normal((ctx, width), 0.01)
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        sd[p + "ln_1.weight"] = torch.ones(width, device=device)
        sd[p + "ln_1.bias"] = torch.zeros(width, device=device)
        sd[p + "attn.in_proj_weight"] = normal((3 * width, width), attn_std)
        sd[p + "attn.in_proj_bias"] = torch.zeros(3 * width, device=device)
        sd[p + "attn.out_proj.weight"] = normal((width, width), proj_std)
        sd[p + "attn.out_proj.bias"] = torch.zeros(width, device=device)
        sd[p + "ln_2.weight"] = torch.ones(width, device=device)
        sd[p + "ln_2.bias"] = torch.zeros(width, device=device)
        sd[p + "mlp.c_fc.weight"] = normal((4 * width, width), fc_std)
        sd[p + "mlp.c_fc.bias"] = normal((4 * width,), 0.01)
        sd[p + "mlp.c_proj.weight"] = normal((width, 4 * width), proj_std)
        sd[p + "mlp.c_proj.bias"] = normal((width,), 0.01)
    sd["ln_final.weight"] = torch.ones(width, device=device)
    sd["ln_final.bias"] = torch.zeros(width, device=device)
    sd["text_projection"] = normal((width, embed_dim), width ** -0.5)
    return sd


def synthetic_token_ids(B, ctx=77, vocab=49408, seed=1, min_len=5, max_len=30):
    """[SOT, n random ids, EOT, 0...] with n ~ U{min_len..max_len}."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.zeros(B, ctx, dtype=torch.int32)
    n = torch.randint(min_len, max_len + 1, (B,), generator=g)
    for b in range(B):
        nb = int(n[b])
        ids[b, 0] = vocab - 2
        ids[b, 1:1 + nb] = torch.randint(1, vocab - 2, (nb,), generator=g, dtype=torch.int32)
        ids[b, 1 + nb] = vocab - 1
    return ids
