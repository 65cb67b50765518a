"""Golden vectors for the BLIP image encoder (blip4cir/vit.py), captured by importing the reference module in the
build container (needs /root/reference).  Run in its own interpreter:

    python tests/golden/make_golden_blipvit.py

vit.py's module-level imports (vit.py:15-20) need timm and fairscale, which are absent offline.  The symbols they
bind (`_cfg`, `PatchEmbed`, `register_model`, `trunc_normal_`, `DropPath`, `named_apply`, `adapt_input_conv`,
`checkpoint_wrapper`) are installed as IMPORT-ONLY stubs that raise if anything ever calls them; with those in place
the reference's own `Mlp`, `Attention`, `Block` (vit.py:23-112) and `VisionTransformer.forward` (vit.py:183-197) run
exactly as written.  What runs from the reference:

  * block level: `Block(dim, heads, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))` on seeded inputs,
    for two shapes (128-wide / 2 heads / 17 tokens and 192-wide / 3 heads / 26 tokens);
  * tower level: a `VisionTransformer` object assembled WITHOUT its __init__ (which would call timm's PatchEmbed and
    trunc_normal_): cls_token, pos_embed, pos_drop, a ModuleList of reference Blocks and the final norm are set as
    attributes, and `patch_embed` is an nn.Module that applies a stride = kernel Conv2d and `flatten(2).transpose(1, 2)`
    - timm's PatchEmbed restated BY READING (timm is not in /root/reference; that one line stays unpinned and is said
    so in DESIGN.md).  `forward(image)` is then the reference's own code: cls concat, + pos_embed, blocks, final norm.
  * `img_embed(..., return_pool_and_normalized=True)` (blip_cir.py:54-70) is restated with torch ops on the tower's
    output: normalize(vision_proj(tokens[:, 0])) - blip_cir.py itself needs transformers' BertTokenizer files.

blip_vit.npz: state-dict in the reference's key names (visual_encoder.*, vision_proj.*), image [3,3,64,64], tokens
[3,17,128], pooled [3,64]; per-block inputs / outputs for the block-level cases."""
import importlib.util
import os
import sys
import types
from functools import partial

import numpy as np
import torch
from torch import nn

OUT = os.path.dirname(os.path.abspath(__file__))


def _never(name):
    def f(*a, **k):
        raise RuntimeError(f"import-only stub {name} was executed: the capture would not be the reference's code")
    return f


def load_vit():
    mods = {}
    for n in ("timm", "timm.models", "timm.models.vision_transformer", "timm.models.registry", "timm.models.layers",
              "timm.models.helpers", "fairscale", "fairscale.nn", "fairscale.nn.checkpoint",
              "fairscale.nn.checkpoint.checkpoint_activations"):
        mods[n] = types.ModuleType(n)
    vt = mods["timm.models.vision_transformer"]
    vt._cfg, vt.PatchEmbed = _never("_cfg"), _never("PatchEmbed")
    mods["timm.models.registry"].register_model = lambda f: f          # decorator position only (module import time)
    mods["timm.models.layers"].trunc_normal_ = _never("trunc_normal_")
    mods["timm.models.layers"].DropPath = _never("DropPath")
    mods["timm.models.helpers"].named_apply = _never("named_apply")
    mods["timm.models.helpers"].adapt_input_conv = _never("adapt_input_conv")
    mods["fairscale.nn.checkpoint.checkpoint_activations"].checkpoint_wrapper = _never("checkpoint_wrapper")
    saved = {n: sys.modules.get(n) for n in mods}
    sys.modules.update(mods)
    try:
        spec = importlib.util.spec_from_file_location("ref_blip_vit", "/root/reference/blip4cir/vit.py")
        vit = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(vit)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    return vit


class PatchEmbedByReading(nn.Module):
    """timm.models.vision_transformer.PatchEmbed restated by reading (not reference code): Conv2d(3, W, kernel = stride =
    patch) -> flatten(2).transpose(1, 2), no norm."""

    def __init__(self, patch, width):
        super().__init__()
        self.proj = nn.Conv2d(3, width, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


def make_block(vit, dim, heads, gen):
    blk = vit.Block(dim=dim, num_heads=heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    with torch.no_grad():                       # seeded, non-trivial parameters (LN affine and biases included)
        for n, p in blk.named_parameters():
            if n.endswith("norm1.weight") or n.endswith("norm2.weight"):
                p.copy_(1 + 0.05 * torch.randn(p.shape, generator=gen))
            elif p.dim() == 1:
                p.copy_(0.05 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
    return blk.eval()


def main():
    vit = load_vit()
    out = {}
    gen = torch.Generator().manual_seed(0)
    # ---- block level (vit.py:91-112 with Attention :46-88 and Mlp :23-43)
    for tag, (dim, heads, B, N) in {"blkA": (128, 2, 3, 17), "blkB": (192, 3, 2, 26)}.items():
        blk = make_block(vit, dim, heads, gen)
        x = torch.randn(B, N, dim, generator=gen)
        with torch.no_grad():
            y = blk(x)
            a = blk.attn(blk.norm1(x))
        out[f"{tag}.x"], out[f"{tag}.y"], out[f"{tag}.attn_out"] = x.numpy(), y.numpy(), a.numpy()
        out[f"{tag}.heads"] = np.int64(heads)
        for k, v in blk.state_dict().items():
            out[f"{tag}.sd.{k}"] = v.numpy()
    # ---- tower level: VisionTransformer.forward (vit.py:183-197) on a model assembled without __init__
    W, layers, heads, patch, res, proj = 128, 2, 2, 16, 64, 64
    S = (res // patch) ** 2 + 1
    model = vit.VisionTransformer.__new__(vit.VisionTransformer)
    nn.Module.__init__(model)
    model.patch_embed = PatchEmbedByReading(patch, W)
    model.cls_token = nn.Parameter(0.05 * torch.randn(1, 1, W, generator=gen))
    model.pos_embed = nn.Parameter(0.05 * torch.randn(1, S, W, generator=gen))
    model.pos_drop = nn.Dropout(p=0.0)
    model.blocks = nn.ModuleList([make_block(vit, W, heads, gen) for _ in range(layers)])
    model.norm = nn.LayerNorm(W, eps=1e-6)
    with torch.no_grad():
        model.norm.weight.copy_(1 + 0.05 * torch.randn(W, generator=gen))
        model.norm.bias.copy_(0.05 * torch.randn(W, generator=gen))
        model.patch_embed.proj.weight.copy_(0.05 * torch.randn(W, 3, patch, patch, generator=gen))
        model.patch_embed.proj.bias.copy_(0.05 * torch.randn(W, generator=gen))
    model.eval()
    vision_proj = nn.Linear(W, proj)
    with torch.no_grad():
        vision_proj.weight.copy_(0.1 * torch.randn(proj, W, generator=gen))
        vision_proj.bias.copy_(0.05 * torch.randn(proj, generator=gen))
    image = torch.randn(3, 3, res, res, generator=gen)
    with torch.no_grad():
        tokens = model(image)                                                        # the reference's forward
        pooled = torch.nn.functional.normalize(vision_proj(tokens[:, 0, :]), dim=-1)  # blip_cir.py:62, restated
    out["image"], out["tokens"], out["pooled"] = image.numpy(), tokens.numpy(), pooled.numpy()
    out["heads"], out["patch"], out["res"] = np.int64(heads), np.int64(patch), np.int64(res)
    for k, v in model.state_dict().items():
        out[f"sd.visual_encoder.{k}"] = v.numpy()
    for k, v in vision_proj.state_dict().items():
        out[f"sd.vision_proj.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "blip_vit.npz"), **out)
    print("wrote blip_vit.npz:", {k: v.shape for k, v in out.items() if not k.startswith(("sd.", "blkA.sd", "blkB.sd"))})


if __name__ == "__main__":
    main()
