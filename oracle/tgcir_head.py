"""Oracle (test infrastructure only): TG-CIR second-stage query producer, CPU fp32/fp64 restatement.

Follows tgcir/models.py: SpatialAttention / TokenLearner (:21-49), Backbone.extract_text_fea (:127-151),
CIRPlus.img_txt_fusion (:198-205), bank_large_step / infonce_loss (:272-296).  Pinned to vectors captured from the
reference classes on CPU (tests/golden/make_golden_tgcir.py -> tests/golden/tgcir_step.npz).
"""
import torch
import torch.nn.functional as F

from . import clip_text

HEAD_KEYS = ("text_fc.weight", "text_fc.bias", "tokenlearn_text.weight", "tokenlearn_text.bias", "masks_text.weight",
             "s_remain_map.0.weight", "s_remain_map.0.bias", "s_remain_map.2.weight", "s_remain_map.2.bias")


def synthetic_head(C=512, S=8, G=4, seed=11):
    """Seeded head parameters (CPU generator: bit-identical on every machine).  tokenlearn_text.weight [S, C] stacks
    the S Conv1d(C, 1, 1) kernels of TokenLearner.tokenizers[s].conv[0]; masks_text as the reference initialises it
    (0.1 everywhere, 1 on the i-th block, models.py:63-69) plus noise so that relu() has both branches."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = lambda shape, std: torch.randn(shape, generator=g) * std
    p = {}
    p["text_fc.weight"] = n((C, C), C ** -0.5)
    p["text_fc.bias"] = n((C,), 0.02)
    p["tokenlearn_text.weight"] = n((S, C), C ** -0.5)
    p["tokenlearn_text.bias"] = n((S,), 0.1)
    m = torch.full((G, C), 0.1)
    for i in range(G):
        m[i, i * (C // G):(i + 1) * (C // G)] = 1.0
    p["masks_text.weight"] = m + n((G, C), 0.2)
    p["s_remain_map.0.weight"] = n((C, 2 * C), (2 * C) ** -0.5)
    p["s_remain_map.0.bias"] = n((C,), 0.02)
    p["s_remain_map.2.weight"] = n((1, C), C ** -0.5)
    p["s_remain_map.2.bias"] = n((1,), 0.1)
    return p


def token_learner(z, w, b):
    """models.py:21-49: per head s, weight_map = sigmoid(conv1d_s(z^T)) [B, 1, L]; out_s = mean_l (z * weight_map)."""
    a = torch.sigmoid(z @ w.t() + b)                    # [B, L, S]
    return torch.einsum("bls,blc->bsc", a, z) / z.shape[1]


def extract_text_fea(tokens, global_fea, head):
    """models.py:127-151 after the transformer: tokens = ln_final(x) [B, L, C], global_fea = pooled @ text_projection."""
    g = global_fea.unsqueeze(1) * F.relu(head["masks_text.weight"]).unsqueeze(0)          # [B, G, C]
    z = tokens @ head["text_fc.weight"].t() + head["text_fc.bias"]
    loc = token_learner(z, head["tokenlearn_text.weight"], head["tokenlearn_text.bias"])
    return torch.cat([g, loc], dim=1)


def img_txt_fusion(ref_token, mod_token, head):
    """models.py:198-205."""
    x = torch.cat([ref_token, mod_token], dim=-1)
    h = F.relu(x @ head["s_remain_map.0.weight"].t() + head["s_remain_map.0.bias"])
    remain = torch.sigmoid(h @ head["s_remain_map.2.weight"].t() + head["s_remain_map.2.bias"])     # [B, NT, 1]
    fuse = remain * ref_token + (1 - remain) * mod_token
    return F.normalize(fuse.mean(dim=1), p=2, dim=-1)


def text_tokens(sd, ids):
    """CLIP text tower up to ln_final over every position + the pooled, projected feature (models.py:128-137)."""
    feats, hidden = clip_text.encode_text(sd, ids, return_hidden=True)
    W = sd["ln_final.weight"].shape[0]
    return F.layer_norm(hidden[-1], (W,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5), feats


def bank_step(sd, head, ids, ref_token, target_bank, labels, tau):
    """models.py:272-296 (bank_large_step + infonce_loss): scalar loss."""
    tokens, feats = text_tokens(sd, ids)
    q = img_txt_fusion(ref_token, extract_text_fea(tokens, feats, head), head)
    return F.cross_entropy((q @ target_bank.t()) / tau, labels.long()), q


# ------------------------------------------------------------------------------------- image side (frozen in stage 2)
IMG_HEAD_KEYS = ("fc.weight", "fc.bias", "tokenlearn.weight", "tokenlearn.bias", "masks.weight")


def synthetic_img_head(C=512, Wv=768, S=8, G=4, seed=13):
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = lambda shape, std: torch.randn(shape, generator=g) * std
    m = torch.full((G, C), 0.1)
    for i in range(G):
        m[i, i * (C // G):(i + 1) * (C // G)] = 1.0
    return {"fc.weight": n((C, Wv), Wv ** -0.5), "fc.bias": n((C,), 0.02), "tokenlearn.weight": n((S, C), C ** -0.5),
            "tokenlearn.bias": n((S,), 0.1), "masks.weight": m + n((G, C), 0.2)}


def extract_img_fea(vis_tokens, global_fea, ihead):
    """Backbone.extract_img_fea (tgcir/models.py:84-125) after the ViT: vis_tokens = transformer output of every
    token (class token included, before ln_post) [B, S, Wv]; global_fea = ln_post(cls) @ proj [B, C]."""
    g = global_fea.unsqueeze(1) * F.relu(ihead["masks.weight"]).unsqueeze(0)
    z = vis_tokens @ ihead["fc.weight"].t() + ihead["fc.bias"]
    loc = token_learner(z, ihead["tokenlearn.weight"], ihead["tokenlearn.bias"])
    return torch.cat([g, loc], dim=1)


def img_embed(vsd, ihead, image):
    """CIRPlus.img_embed(image, return_pool_and_normalized=True) (models.py:183-196): (tokens [B, 12, C], pooled)."""
    from . import clip_vision
    feats, tokens = clip_vision.encode_image(vsd, image, return_tokens=True)
    emb = extract_img_fea(tokens, feats, ihead)
    return emb, F.normalize(emb.mean(dim=1), p=2, dim=-1)
