"""Seeded synthetic weights / inputs for benchmarks and smoke runs.

No pretrained CLIP weights or datasets exist offline, so the bench uses random weights with
the distributions of CLIP.initialize_parameters (clip4cir/clip/model.py:301-328) and
FashionIQ-shaped synthetic triplets.  Everything is drawn from the torch CPU generator so the
values are bit-identical on every machine."""
import torch

CLIP_TEXT_CONFIGS = {
    # name: (width, layers, heads, embed_dim)   (clip/model.py:420-426 applied to the OpenAI checkpoints)
    "ViT-B/32": (512, 12, 8, 512),
    "ViT-B/16": (512, 12, 8, 512),
    "ViT-L/14": (768, 12, 12, 768),
    "RN50x4": (640, 12, 10, 640),
    "RN50": (512, 12, 8, 1024),
    "RN101": (512, 12, 8, 512),
    "RN50x16": (768, 12, 12, 768),
    "RN50x64": (1024, 12, 16, 1024),
    "ViT-L/14@336px": (768, 12, 12, 768),
}


def text_state_dict(width, layers, embed_dim, vocab=49408, ctx=77, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)

    def normal(shape, std):
        return torch.randn(shape, generator=g) * std

    sd = {"token_embedding.weight": normal((vocab, width), 0.02), "positional_embedding": normal((ctx, width), 0.01)}
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        sd[p + "ln_1.weight"] = torch.ones(width)
        sd[p + "ln_1.bias"] = torch.zeros(width)
        sd[p + "attn.in_proj_weight"] = normal((3 * width, width), attn_std)
        sd[p + "attn.in_proj_bias"] = torch.zeros(3 * width)
        sd[p + "attn.out_proj.weight"] = normal((width, width), proj_std)
        sd[p + "attn.out_proj.bias"] = torch.zeros(width)
        sd[p + "ln_2.weight"] = torch.ones(width)
        sd[p + "ln_2.bias"] = torch.zeros(width)
        sd[p + "mlp.c_fc.weight"] = normal((4 * width, width), fc_std)
        sd[p + "mlp.c_fc.bias"] = normal((4 * width,), 0.01)
        sd[p + "mlp.c_proj.weight"] = normal((width, 4 * width), proj_std)
        sd[p + "mlp.c_proj.bias"] = normal((width,), 0.01)
    sd["ln_final.weight"] = torch.ones(width)
    sd["ln_final.bias"] = torch.zeros(width)
    sd["text_projection"] = normal((width, embed_dim), width ** -0.5)
    return sd


def token_ids(B, ctx=77, vocab=49408, seed=1, min_len=5, max_len=30):
    """[SOT, n random ids, EOT, 0...], n ~ U{min_len..max_len}; EOT = vocab-1 is the row maximum."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.zeros(B, ctx, dtype=torch.int32)
    n = torch.randint(min_len, max_len + 1, (B,), generator=g)
    for b in range(B):
        nb = int(n[b])
        ids[b, 0] = vocab - 2
        ids[b, 1:1 + nb] = torch.randint(1, vocab - 2, (nb,), generator=g, dtype=torch.int32)
        ids[b, 1 + nb] = vocab - 1
    return ids


def banks(M, D, seed=2):
    """(target_bank = normalize(randn(M, D)), refer_bank = randn(M, D)), seeds 2 and 3."""
    g2 = torch.Generator(device="cpu").manual_seed(seed)
    g3 = torch.Generator(device="cpu").manual_seed(seed + 1)
    target = torch.nn.functional.normalize(torch.randn(M, D, generator=g2))
    refer = torch.randn(M, D, generator=g3)
    return target, refer


def triplet_indices(B, M, seed=4):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(0, M, (B,), generator=g), torch.randint(0, M, (B,), generator=g)
