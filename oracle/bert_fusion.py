"""Oracle: BLIP multimodal text encoder (CPU fp32; torch autograd gives the backward).

Restates blip4cir/med.py: BertEmbeddings (:68-112: word + absolute position, LayerNorm eps 1e-12, no
token-type embeddings), BertSelfAttention (:115-243: separate query/key/value Linear, scores / sqrt(64)
+ additive mask, softmax), BertSelfOutput / BertOutput (:246-257,324-335: dense + residual + post-LN),
BertIntermediate (:309-321: exact GELU), BertLayer in mode='multimodal' (:351-397: self-attention ->
cross-attention over the image tokens -> FFN), the (1-mask)*-10000 extended mask (:686), and
blip_cir.py:98 (normalize(text_proj(h[:,0]))).  Dropout is inactive (blip4cir/train.py:111 calls
model.blip.eval()).  Parameters: a dict keyed like BertModel.state_dict() (+ text_proj.*)."""
import math

import torch
import torch.nn.functional as F


def cfg_from_state_dict(sd):
    W = sd["embeddings.word_embeddings.weight"].shape[1]
    layers = len({k.split(".")[2] for k in sd if k.startswith("encoder.layer.")})
    return {"hidden": W, "heads": W // 64, "layers": layers,
            "vocab": sd["embeddings.word_embeddings.weight"].shape[0],
            "max_pos": sd["embeddings.position_embeddings.weight"].shape[0],
            "intermediate": sd["encoder.layer.0.intermediate.dense.weight"].shape[0],
            "enc_width": sd["encoder.layer.0.crossattention.self.key.weight"].shape[1],
            "proj_dim": sd["text_proj.weight"].shape[0] if "text_proj.weight" in sd else None}


def _lin(sd, p, x):
    return x @ sd[p + ".weight"].t() + sd[p + ".bias"]


def _attn(sd, p, x, kv, bias, heads):
    """BertSelfAttention + BertSelfOutput: x [B,Lq,W] queries, kv [B,Lk,*] keys/values source."""
    B, Lq, W = x.shape
    Lk = kv.shape[1]
    hd = W // heads
    q = _lin(sd, p + ".self.query", x).view(B, Lq, heads, hd).transpose(1, 2)
    k = _lin(sd, p + ".self.key", kv).view(B, Lk, heads, hd).transpose(1, 2)
    v = _lin(sd, p + ".self.value", kv).view(B, Lk, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if bias is not None:
        s = s + bias[:, None, None, :]
    ctx = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, W)
    y = _lin(sd, p + ".output.dense", ctx) + x
    return F.layer_norm(y, (W,), sd[p + ".output.LayerNorm.weight"], sd[p + ".output.LayerNorm.bias"], 1e-12)


def fusion_forward(sd, ids, mask, enc):
    """ids [B,L] int, mask [B,L] {0,1}, enc [B,S,enc_width] -> last_hidden_state [B,L,W]."""
    cfg = cfg_from_state_dict(sd)
    W = cfg["hidden"]
    ids = ids.long()
    L = ids.shape[1]
    x = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.position_embeddings.weight"][:L]
    x = F.layer_norm(x, (W,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], 1e-12)
    bias = (1.0 - mask.to(x.dtype)) * -10000.0
    for l in range(cfg["layers"]):
        p = f"encoder.layer.{l}"
        x = _attn(sd, p + ".attention", x, x, bias, cfg["heads"])
        x = _attn(sd, p + ".crossattention", x, enc, None, cfg["heads"])      # image mask is all ones
        u = F.gelu(_lin(sd, p + ".intermediate.dense", x))
        y = _lin(sd, p + ".output.dense", u) + x
        x = F.layer_norm(y, (W,), sd[p + ".output.LayerNorm.weight"], sd[p + ".output.LayerNorm.bias"], 1e-12)
    return x


def fusion_query(sd, ids, mask, enc):
    """blip_cir.py:98: L2-normalised text_proj of the [ENC] position."""
    h = fusion_forward(sd, ids, mask, enc)
    return F.normalize(_lin(sd, "text_proj", h[:, 0, :]), dim=-1)
