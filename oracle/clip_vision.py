"""Oracle: CLIP VisionTransformer forward (CPU fp32).

Restates clip4cir/clip/model.py:206-242 (VisionTransformer): bias-free patch-embedding
convolution with stride == kernel (an im2col GEMM), class token, positional embedding,
ln_pre, un-masked pre-LN residual blocks (same block as the text tower, :171-192),
ln_post on the class token and the ``x @ proj`` projection.
"""
import torch
import torch.nn.functional as F
from .clip_text import residual_block


def vision_cfg_from_state_dict(sd):
    """clip4cir/clip/model.py:404-411."""
    width = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    return {"width": width, "patch": patch, "grid": grid, "resolution": patch * grid,
            "layers": layers, "heads": width // 64, "embed_dim": sd["visual.proj"].shape[1]}


def patchify(image, patch):
    """[B,3,H,W] -> [B, grid*grid, 3*patch*patch] with the (c, ky, kx) ordering of a conv weight."""
    B, C, H, W = image.shape
    gh, gw = H // patch, W // patch
    x = image.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, gh * gw, C * patch * patch)


def encode_image(sd, image, return_tokens=False):
    """return_tokens: also the transformer output of EVERY token before ln_post (what TG-CIR's
    Backbone.extract_img_fea feeds to its fc + TokenLearner, tgcir/models.py:84-125)."""
    cfg = vision_cfg_from_state_dict(sd)
    W = cfg["width"]
    B = image.shape[0]
    w = sd["visual.conv1.weight"].reshape(W, -1)                  # [W, 3*p*p]
    x = patchify(image.float(), cfg["patch"]) @ w.t()             # conv1, model.py:224-226
    cls = sd["visual.class_embedding"].expand(B, 1, W)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]   # :227-230
    x = F.layer_norm(x, (W,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    for i in range(cfg["layers"]):
        x = residual_block(x, sd, f"visual.transformer.resblocks.{i}.", cfg["heads"], None)
    tokens = x
    x = F.layer_norm(x[:, 0, :], (W,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    feats = x @ sd["visual.proj"]                                 # :237-240
    return (feats, tokens) if return_tokens else feats


def synthetic_vision_state_dict(width, layers, patch, res, embed_dim, seed=0):
    """Seeded random ViT weights under CLIP's `visual.*` keys (CPU generator: bit-identical on every machine)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = lambda shape, std: torch.randn(shape, generator=g) * std
    S = (res // patch) ** 2 + 1
    sd = {"visual.conv1.weight": n((width, 3, patch, patch), (3 * patch * patch) ** -0.5),
          "visual.class_embedding": n((width,), width ** -0.5),
          "visual.positional_embedding": n((S, width), width ** -0.5),
          "visual.ln_pre.weight": 1 + n((width,), 0.05), "visual.ln_pre.bias": n((width,), 0.05),
          "visual.ln_post.weight": 1 + n((width,), 0.05), "visual.ln_post.bias": n((width,), 0.05),
          "visual.proj": n((width, embed_dim), width ** -0.5)}
    for i in range(layers):
        p = f"visual.transformer.resblocks.{i}."
        sd[p + "ln_1.weight"] = 1 + n((width,), 0.05); sd[p + "ln_1.bias"] = n((width,), 0.05)
        sd[p + "attn.in_proj_weight"] = n((3 * width, width), width ** -0.5)
        sd[p + "attn.in_proj_bias"] = n((3 * width,), 0.02)
        sd[p + "attn.out_proj.weight"] = n((width, width), width ** -0.5 * (2 * layers) ** -0.5)
        sd[p + "attn.out_proj.bias"] = n((width,), 0.02)
        sd[p + "ln_2.weight"] = 1 + n((width,), 0.05); sd[p + "ln_2.bias"] = n((width,), 0.05)
        sd[p + "mlp.c_fc.weight"] = n((4 * width, width), (2 * width) ** -0.5)
        sd[p + "mlp.c_fc.bias"] = n((4 * width,), 0.02)
        sd[p + "mlp.c_proj.weight"] = n((width, 4 * width), width ** -0.5 * (2 * layers) ** -0.5)
        sd[p + "mlp.c_proj.bias"] = n((width,), 0.02)
    return sd
