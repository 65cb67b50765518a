"""Seeded input builders shared by make_golden.py (capture) and the tests (replay).

Inputs that are too big to commit (the 4 099x512 and 40 000x768 banks) are regenerated
from the torch CPU generator, which is bit-identical across machines; only the reference's
OUTPUTS for them are stored in loss_cases.npz."""
import torch

LOSS_CASES = [(4, 500, 64, 0.01), (32, 4099, 512, 0.02), (16, 40000, 768, 0.03)]


def loss_case_inputs(ci):
    b, m, d, tau = LOSS_CASES[ci]
    g = torch.Generator().manual_seed(100 + ci)
    text = torch.randn(b, d, generator=g)
    refer_bank = torch.randn(m, d, generator=g)
    bank = torch.nn.functional.normalize(torch.randn(m, d, generator=g))
    ridx = torch.randint(0, m, (b,), generator=g)
    labels = torch.randint(0, m, (b,), generator=g)
    if ci == 1:   # label at the first row, the last row, and a duplicate label
        labels[0], labels[1], labels[2] = 0, m - 1, labels[3]
    return text, refer_bank, bank, ridx, labels, tau


# ----------------------------------------------------------------------------- TG-CIR step (tgcir_step.npz)
# Weights come from oracle.clip_text.synthetic_text_state_dict(C, LAYERS, C, VOCAB, L, seed=7) and
# oracle.tgcir_head.synthetic_head(C, 8, 4, seed=11); large gradients are stored as every SAMPLE-th element + L2 norm.
TGCIR = dict(B=6, L=77, C=512, VOCAB=1024, LAYERS=2, M=300, TAU=0.02, SAMPLE=97, TEXT_SEED=7, HEAD_SEED=11)


def tgcir_inputs():
    from oracle import clip_text
    t = TGCIR
    g = torch.Generator().manual_seed(31)
    ids = clip_text.synthetic_token_ids(t["B"], ctx=t["L"], vocab=t["VOCAB"], seed=5, min_len=3, max_len=20)
    ref = torch.randn(t["B"], 12, t["C"], generator=g) * 0.5
    bank = torch.nn.functional.normalize(torch.randn(t["M"], t["C"], generator=g), dim=-1)
    labels = torch.randint(0, t["M"], (t["B"],), generator=g)
    return ids, ref, bank, labels


def tgcir_weights():
    from oracle import clip_text, tgcir_head
    t = TGCIR
    sd = clip_text.synthetic_text_state_dict(t["C"], t["LAYERS"], t["C"], vocab=t["VOCAB"], ctx=t["L"], seed=t["TEXT_SEED"])
    return sd, tgcir_head.synthetic_head(t["C"], 8, 4, seed=t["HEAD_SEED"])


def tgcir_image_side():
    """Seeded vision tower (res 32, patch 16, width 768 - TG-CIR's fc is Linear(768, 512)), image-side head, images."""
    from oracle import clip_vision, tgcir_head
    vsd = clip_vision.synthetic_vision_state_dict(768, 2, 16, 32, TGCIR["C"], seed=17)
    ihead = tgcir_head.synthetic_img_head(TGCIR["C"], 768, 8, 4, seed=13)
    images = torch.randn(5, 3, 32, 32, generator=torch.Generator().manual_seed(19))
    return vsd, ihead, images


def tgcir_grad_check(z, name, g, tol):
    """Compare a gradient with its stored summary: L2 norm + every SAMPLE-th element (all of it when small)."""
    g = g.detach().reshape(-1).double().cpu()
    ref = torch.from_numpy(z["grad::" + name]).double()
    got = g if g.numel() <= 8192 else g[::TGCIR["SAMPLE"]]
    gn = float(z["gnorm::" + name])
    assert abs(g.norm().item() - gn) <= tol * gn + 1e-12, (name, g.norm().item(), gn)
    floor = gn * (got.numel() / max(1, g.numel())) ** 0.5          # expected norm of the sample
    err = (got - ref).norm().item()
    assert err <= tol * max(ref.norm().item(), floor) + 1e-12, (name, err, ref.norm().item())


# ----------------------------------------------------------------------- fusion-style validation (valfusion.npz)
def valfusion_inputs():
    """Synthetic token gallery [NG, 12, 512] (the reference hard-codes the 512-wide feature), its pooled + normalised
    form, stub query features and the FashionIQ / CIRR rows; regenerated from seeds on both sides, only the metrics are
    stored."""
    g = torch.Generator().manual_seed(23)
    D, NG, NQ, T = 512, 260, 90, 12
    tokens = torch.randn(NG, T, D, generator=g)
    pooled = torch.nn.functional.normalize(tokens.mean(dim=1), dim=-1)
    names = [f"img{i:04d}" for i in range(NG)]
    ref_i = torch.randint(0, NG, (NQ,), generator=g)
    tgt_i = (ref_i + 1 + torch.randint(0, NG - 1, (NQ,), generator=g)) % NG
    q = 0.045 * torch.randn(NQ, D, generator=g) + pooled[tgt_i] * torch.rand(NQ, 1, generator=g) * 0.6
    fiq_rows = [(names[int(r)], names[int(t)], [f"cap a {i}.", f"cap b {i}?"]) for i, (r, t) in enumerate(zip(ref_i, tgt_i))]
    members = []
    for r, t in zip(ref_i, tgt_i):
        pool = [int(x) for x in torch.randperm(NG, generator=g)[:12] if int(x) not in (int(r), int(t))][:5]
        members.append([names[j] for j in pool + [int(t)]])
    cirr_rows = [(names[int(r)], names[int(t)], f"cap {i}", members[i]) for i, (r, t) in enumerate(zip(ref_i, tgt_i))]
    return tokens, pooled, names, q, fiq_rows, cirr_rows


class StubFusion:
    """img_txt_fusion returns pre-made query rows in call order, nudged by the gathered reference tokens (so a wrong
    gather shows)."""

    def __init__(self, q):
        self.q, self.pos = q, 0

    def img_txt_fusion(self, ref_token, mod):
        n = len(mod)
        assert ref_token.shape[0] == n and ref_token.dim() == 3
        out = self.q[self.pos:self.pos + n].to(ref_token.device) + 0.05 * ref_token.mean(dim=1)
        self.pos += n
        return out


# ----------------------------------------------------------------------------- BLIP-2 stage-2 loss (blip2_stage2.npz)
BLIP2_CASES = {"small": dict(B=6, M=37, H=96, D=256, L=8, seed=0, ties=True),
               "m1000": dict(B=16, M=1000, H=96, D=256, L=8, seed=1, ties=False)}


def blip2_target_feats(tag):
    """The static token bank [M, 32, D] of a case (too big to store for M = 1000): its own seeded generator; `ties` plants
    equal token rows inside two targets (several arg-max rows: torch.max takes the first)."""
    c = BLIP2_CASES[tag]
    g = torch.Generator().manual_seed(1000 + c["seed"])
    t = torch.nn.functional.normalize(torch.randn(c["M"], 32, c["D"], generator=g), dim=-1)
    if c["ties"]:
        t[3, 5] = t[3, 17]
        t[7, :] = t[7, 0]
    return t


# ------------------------------------------------- blip4cir end to end: strings in, metrics out (blip_val.npz)
# Sizes the reference hard-codes (blip4cir/utils.py:36-37, validate.py:83): 577 image tokens x 768, 256-wide pooled features.
BLIPVAL = dict(W=768, VLAYERS=1, PATCH=16, RES=384, PROJ=256, HID=128, LAYERS=2, INTER=512, MAXPOS=128, NG=20, NQ=24, B=6, M=20,
               TAU=0.03)
_BLIP_WORDS = ("is are has with and more less very dress shirt top skirt sleeve sleeves collar floral print striped solid red blue "
               "green black white pink dark light long short longer shorter loose fitted plain casual same different color style "
               "the a dog dogs cat people remove add change instead facing left right two").split()


def blipval_vocab():
    """The synthetic WordPiece vocabulary of make_golden_bert_tokenizer.py (stored in bert_tokenizer.json)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bert_tokenizer.json"), encoding="utf-8") as f:
        return json.load(f)["vocab"]


def blipval_state_dict():
    """BLIP_Retrieval state dict (blip_cir.py:34-47 names) from seeds: 1-block ViT at the full token shape, 2-layer
    cross-attention BERT 128 wide over 768-wide image tokens, vocab = synthetic vocabulary + [DEC] + [ENC]."""
    from oracle import blip_vit
    c = BLIPVAL
    sd = blip_vit.synthetic_state_dict(c["W"], c["VLAYERS"], c["PATCH"], c["RES"], c["PROJ"], seed=41)
    for k in list(sd):                       # 768-wide products: keep the block's activations O(1)
        if k.endswith(("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")):
            sd[k] = sd[k] * 0.4
    g = torch.Generator().manual_seed(43)
    r = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    H, E, I, V = c["HID"], c["W"], c["INTER"], len(blipval_vocab()) + 2
    t = {"embeddings.word_embeddings.weight": r(V, H, std=0.5), "embeddings.position_embeddings.weight": r(c["MAXPOS"], H, std=0.1),
         "embeddings.LayerNorm.weight": 1 + r(H, std=0.1), "embeddings.LayerNorm.bias": r(H, std=0.05)}
    for l in range(c["LAYERS"]):
        p = f"encoder.layer.{l}."
        for a, kw in (("attention", H), ("crossattention", E)):
            for n in ("query", "key", "value"):
                w_in = H if n == "query" else kw
                t[p + f"{a}.self.{n}.weight"] = r(H, w_in, std=0.6 / w_in ** 0.5)
                t[p + f"{a}.self.{n}.bias"] = r(H, std=0.05)
            t[p + a + ".output.dense.weight"] = r(H, H, std=0.06)
            t[p + a + ".output.dense.bias"] = r(H, std=0.05)
            t[p + a + ".output.LayerNorm.weight"] = 1 + r(H, std=0.1)
            t[p + a + ".output.LayerNorm.bias"] = r(H, std=0.05)
        t[p + "intermediate.dense.weight"] = r(I, H, std=0.06); t[p + "intermediate.dense.bias"] = r(I, std=0.05)
        t[p + "output.dense.weight"] = r(H, I, std=0.04); t[p + "output.dense.bias"] = r(H, std=0.05)
        t[p + "output.LayerNorm.weight"] = 1 + r(H, std=0.1); t[p + "output.LayerNorm.bias"] = r(H, std=0.05)
    sd.update({"text_encoder." + k: v for k, v in t.items()})
    sd["text_proj.weight"], sd["text_proj.bias"] = r(c["PROJ"], H, std=0.08), r(c["PROJ"], std=0.05)
    return sd


def blipval_inputs():
    """-> dict(images [NG,3,384,384], names, fiq_rows, cirr_rows, train = (captions, indexs, target_ids, refer_ids))."""
    import random
    c = BLIPVAL
    g = torch.Generator().manual_seed(47)
    rng = random.Random(53)
    images = torch.randn(c["NG"], 3, c["RES"], c["RES"], generator=g)
    names = [f"B{i:05d}" for i in range(c["NG"])]
    cap = lambda lo, hi: " ".join(rng.choice(_BLIP_WORDS) for _ in range(rng.randint(lo, hi)))
    ref_i = [rng.randrange(c["NG"]) for _ in range(c["NQ"])]
    tgt_i = [(r + 1 + rng.randrange(c["NG"] - 1)) % c["NG"] for r in ref_i]
    fiq_rows = [(names[r], names[t], [cap(2, 7) + rng.choice(["", ".", "?", " ,"]), cap(2, 6) + rng.choice(["", ".", "!"])])
                for r, t in zip(ref_i, tgt_i)]
    cirr_rows = []
    for r, t in zip(ref_i, tgt_i):
        others = [j for j in rng.sample(range(c["NG"]), 9) if j not in (r, t)][:5]
        cirr_rows.append((names[r], names[t], cap(3, 12).capitalize() + rng.choice(["", ".", ", too"]), [names[j] for j in others + [t]]))
    caps = [cap(3, 10).capitalize() + " and " + cap(2, 6) for _ in range(c["B"])]
    refer = [rng.randrange(c["NG"]) for _ in range(c["B"])]
    target = [rng.randrange(c["M"]) for _ in range(c["B"])]
    train = (caps, torch.arange(c["B"]), torch.tensor(target), torch.tensor(refer))
    return dict(images=images, names=names, fiq_rows=fiq_rows, cirr_rows=cirr_rows, train=train)


class ClassicRows(torch.utils.data.Dataset):
    """'classic' dataset items (name, preprocessed image) as data_utils.py returns them."""
    data_name, split, dress_types = "fiq", "val", ["dress"]

    def __init__(self, names, images):
        self.names, self.images = names, images

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        return self.names[i], self.images[i]


# ------------------------------------------------------------- BLIP fusion encoder at config 4's full shape
def fusion_sd(layers, W, I, E, Dp, vocab, max_pos, seed, init="small_residual"):
    """BertModel(add_cross_attention) + text_proj state-dict with med.py's key names: BertPreTrainedModel's normal init scaled
    up on the query / key / value / intermediate matrices so that attention is not uniform, non-trivial LayerNorm affine, and
    SMALL residual-branch outputs (attention / cross-attention / FFN output.dense at std 0.01).  The last point keeps the
    12-layer post-LN stack from rank-collapsing: with all matrices at std 0.04 the mean pairwise cosine between the positions'
    hidden states grows 0.20, 0.45, 0.67, ... 0.9996, 0.9998 over the layers (every position carries the same vector at the
    top: the near-uniform cross-attention over 577 random image tokens adds one common vector per layer), the query / key
    gradients of the top layers are then the remainder of a cancelling sum and no bf16 attention backward reproduces them to
    better than ~0.2; with this init it ends at 0.47 and every tensor is held to the same gate.
    init="reference": BertPreTrainedModel._init_weights' own scale on every matrix (normal std 0.02, med.py / HF initializer_range;
    LayerNorm affine and biases as above so that they matter) - the rank-collapsing regime, kept so that the noise floor of BOTH
    inits is on record (noise_floor.json: blip_*_refinit)."""
    g = torch.Generator().manual_seed(seed)
    ref_init = init == "reference"
    r = lambda *s, std=0.02: torch.randn(*s, generator=g) * (0.02 if (ref_init and len(s) == 2) else std)
    sd = {"embeddings.word_embeddings.weight": r(vocab, W), "embeddings.position_embeddings.weight": r(max_pos, W),
          "embeddings.LayerNorm.weight": 1 + r(W, std=0.1), "embeddings.LayerNorm.bias": r(W, std=0.05)}
    for l in range(layers):
        p = f"encoder.layer.{l}."
        for a, kw in (("attention", W), ("crossattention", E)):
            sd[p + a + ".self.query.weight"] = r(W, W, std=0.04); sd[p + a + ".self.query.bias"] = r(W, std=0.05)
            sd[p + a + ".self.key.weight"] = r(W, kw, std=0.04); sd[p + a + ".self.key.bias"] = r(W, std=0.05)
            sd[p + a + ".self.value.weight"] = r(W, kw, std=0.04); sd[p + a + ".self.value.bias"] = r(W, std=0.05)
            sd[p + a + ".output.dense.weight"] = r(W, W, std=0.01); sd[p + a + ".output.dense.bias"] = r(W, std=0.05)
            sd[p + a + ".output.LayerNorm.weight"] = 1 + r(W, std=0.1); sd[p + a + ".output.LayerNorm.bias"] = r(W, std=0.05)
        sd[p + "intermediate.dense.weight"] = r(I, W, std=0.04); sd[p + "intermediate.dense.bias"] = r(I, std=0.05)
        sd[p + "output.dense.weight"] = r(W, I, std=0.01); sd[p + "output.dense.bias"] = r(W, std=0.05)
        sd[p + "output.LayerNorm.weight"] = 1 + r(W, std=0.1); sd[p + "output.LayerNorm.bias"] = r(W, std=0.05)
    sd["text_proj.weight"] = r(Dp, W, std=0.05); sd["text_proj.bias"] = r(Dp, std=0.05)
    return sd
