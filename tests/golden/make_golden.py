"""Generate the golden vectors under tests/golden/ by importing the reference itself.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied: the script imports its modules, feeds seeded inputs and stores the
inputs and outputs as arrays.  Import recipe: SURVEY.md Appendix A (stub torchvision/ftfy,
flat sibling imports, early-failing torch.jit.load).

    python tests/golden/make_golden.py

Files written (all .npz, fp32 unless noted):
  tokenizer.npz        captions (json string) -> int32 ids [n,77]             (clip.tokenize)
  tiny_clip.npz        tiny CLIP state-dict (as held by the model after load, i.e. with the
                       fp16 rounding of build_model applied), ids, image -> encode_text,
                       encode_image outputs + per-block hidden states of the text tower
  cirplus_step.npz     CIRPlus.forward bank_loss + gradients of every text-tower parameter
                       (models_negplus, plus=True) and the zscir-style per-triplet variant
  loss_cases.npz       bank InfoNCE loss + dq for several (B, M, D, tau)
  recall.npz           compute_fiq_val_metrics / compute_cirr_val_metrics on synthetic galleries
  adamw.npz            two torch.optim.AdamW steps with the reference's hyper-parameters
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_stubs():
    tv, tvt, tvf = (types.ModuleType(n) for n in
                    ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"))

    class _T:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize"):
        setattr(tvt, n, _T)
    tvt.InterpolationMode = type("IM", (), {"BICUBIC": 3})
    tvf.pad = lambda img, *a, **k: img
    tv.transforms, tvt.functional = tvt, tvf
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf, "ftfy": ftfy})

    def _nojit(*a, **k):
        raise RuntimeError("not a JIT archive")

    torch.jit.load = _nojit


class FakeFiqDataset(torch.utils.data.Dataset):
    """Duck-typed relative-val dataset: (reference_name, target_name, [cap0, cap1])."""
    dress_types = ["dress"]
    data_name = "fiq"
    split = "val"

    def __init__(self, rows):
        self.rows = rows

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return self.rows[i]


class FakeCirrDataset(FakeFiqDataset):
    data_name = "cirr"


class FakeModel:
    """encode_text returns pre-made features in call order; combiner = element-wise sum."""

    def __init__(self, text_feats, output_dim):
        self.text_feats = text_feats
        self.output_dim = output_dim
        self.pos = 0

    def encode_text(self, captions):
        n = len(captions)
        out = self.text_feats[self.pos:self.pos + n]
        self.pos += n
        return out

    def combining_function(self, a, b):
        return a + b


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "clip4cir"))
    import clip                      # noqa: E402
    import models_negplus            # noqa: E402
    import validate                  # noqa: E402
    from clip.model import CLIP      # noqa: E402

    # ---- 1. tokenizer -------------------------------------------------------------------
    captions = [
        "is red and has long sleeves", "is shorter and more colorful", "has a v-neck and is black",
        "is less formal with a floral print", "the dog is sitting on the grass instead of standing",
        "remove the people and add a second bus", "is a darker shade of blue, and sleeveless.",
        "has thinner straps & a higher hemline", "shows two cats looking at the camera?",
        "a", "is solid white with 3/4 sleeves", "more plain and less frilly",
        "is grey with a graphic print and is more casual and has shorter sleeves and a rounder neck",
        "Same breed of dog but it's facing left, with a ball in its mouth",
        "is darker and has longer sleeves and is darker with a belt",
    ]
    ids = clip.tokenize(captions).numpy().astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "tokenizer.npz"), captions=json.dumps(captions), ids=ids)

    # ---- 2. tiny CLIP -------------------------------------------------------------------
    torch.manual_seed(0)
    VOC, CTX, TW, TL, ED = 512, 77, 128, 2, 64
    tiny = CLIP(ED, 32, 2, 128, 16, CTX, VOC, TW, TW // 64, TL)
    with torch.no_grad():   # make LN affine and biases non-trivial so parity checks see them
        for n, p in tiny.named_parameters():
            if n.endswith("ln_1.weight") or n.endswith("ln_2.weight") or "ln_final.weight" in n \
                    or "ln_pre.weight" in n or "ln_post.weight" in n:
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith(".bias") or n.endswith("in_proj_bias"):
                p.add_(0.05 * torch.randn_like(p))
    tmp = "/tmp/_tiny_clip_sd.pt"
    torch.save(tiny.state_dict(), tmp)
    model = models_negplus.CIRPlus(tmp, tau=0.02, device=torch.device("cpu"), plus=True)
    sd = {k: v.detach().clone() for k, v in model.clip.state_dict().items()}

    g = torch.Generator().manual_seed(1)
    B = 6
    tok = torch.zeros(B, CTX, dtype=torch.int32)
    lens = [3, 9, 20, 1, 75, 12]
    for b, n in enumerate(lens):
        tok[b, 0] = VOC - 2
        tok[b, 1:1 + n] = torch.randint(1, VOC - 2, (n,), generator=g, dtype=torch.int32)
        tok[b, 1 + n] = VOC - 1
    image = torch.randn(3, 3, 32, 32, generator=g)

    hidden = []
    hooks = [blk.register_forward_hook(lambda m, i, o: hidden.append(o.detach().permute(1, 0, 2).clone()))
             for blk in model.clip.transformer.resblocks]
    with torch.no_grad():
        text_feats = model.clip.encode_text(tok)
        image_feats = model.clip.encode_image(image)
    for h in hooks:
        h.remove()
    # vision blocks also fired the hook? no: hooks are only on the text transformer
    out = {"sd::" + k: v.numpy() for k, v in sd.items()}
    out.update(ids=tok.numpy(), image=image.numpy(), text_feats=text_feats.numpy(),
               image_feats=image_feats.numpy())
    for i, h in enumerate(hidden[:TL]):
        out[f"hidden_{i}"] = h.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_clip.npz"), **out)

    # ---- 3. CIRPlus.forward loss + grads --------------------------------------------------
    # clip.tokenize would emit ids up to 49407; the tiny model has a 512-row vocabulary, so the
    # tokenizer (pinned separately above) is replaced by the pre-made ids for this capture.
    N_IMG, M_UNL = 40, 25
    refer_bank = torch.randn(N_IMG, ED, generator=g)
    target_bank = torch.nn.functional.normalize(torch.randn(N_IMG + M_UNL, ED, generator=g))
    ref_img_ids = torch.randint(0, N_IMG, (B,), generator=g)
    tgt_img_ids = torch.randint(0, N_IMG, (B,), generator=g)
    trip_idx = torch.arange(B)
    real_tokenize = clip.tokenize
    clip.tokenize = lambda text, *a, **k: tok
    model.refer_bank, model.target_bank = refer_bank.clone(), target_bank.clone()
    model.zero_grad()
    loss = model.forward(["x"] * B, trip_idx, tgt_img_ids, ref_img_ids)["bank_loss"]
    loss.backward()
    step = {"refer_bank": refer_bank.numpy(), "target_bank": target_bank.numpy(),
            "ref_img_ids": ref_img_ids.numpy(), "tgt_img_ids": tgt_img_ids.numpy(),
            "tau": np.float32(0.02), "loss_plus": loss.detach().numpy()}
    for n, p in model.clip.named_parameters():
        if p.grad is not None:
            step["grad_plus::" + n] = p.grad.numpy().copy()
        else:
            assert n.startswith("visual.") or n == "logit_scale", n
    # per-triplet variant (plus=False): reference row = refer_bank[indexs] (models_negplus.py:135)
    model.plus = False
    trip_bank = torch.randn(B, ED, generator=g)
    model.refer_bank = trip_bank.clone()
    model.zero_grad()
    loss2 = model.forward(["x"] * B, trip_idx, tgt_img_ids, ref_img_ids)["bank_loss"]
    loss2.backward()
    step.update(trip_bank=trip_bank.numpy(), loss_trip=loss2.detach().numpy(),
                grad_trip_text_projection=model.clip.text_projection.grad.numpy().copy())
    clip.tokenize = real_tokenize
    np.savez_compressed(os.path.join(OUT, "cirplus_step.npz"), **step)

    # ---- 4. loss-only cases ---------------------------------------------------------------
    sys.path.insert(0, OUT)
    from cases import LOSS_CASES, loss_case_inputs
    cases = {}
    for ci in range(len(LOSS_CASES)):
        text, rb, bank, ridx, lab, tau = loss_case_inputs(ci)
        text.requires_grad_(True)
        model.plus, model.tau = True, tau
        model.refer_bank, model.target_bank = rb, bank
        ld = {}
        model.bank_large_step(ld, text, None, lab, ridx)
        ld["bank_loss"].backward()
        cases[f"c{ci}_loss"] = ld["bank_loss"].detach().numpy()
        cases[f"c{ci}_dtext"] = text.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **cases)

    # ---- 5. Recall@K ----------------------------------------------------------------------
    gr = torch.Generator().manual_seed(7)
    D, NG, NQ = 64, 300, 96
    gallery = torch.randn(NG, D, generator=gr)
    names = [f"img{i:04d}" for i in range(NG)]
    text_f = 0.7 * torch.randn(NQ, D, generator=gr)
    ref_i = torch.randint(0, NG, (NQ,), generator=gr)
    tgt_i = (ref_i + 1 + torch.randint(0, NG - 1, (NQ,), generator=gr)) % NG     # != reference
    # make targets retrievable: pull the query towards the target
    text_f = text_f + 0.35 * gallery[tgt_i] * torch.rand(NQ, 1, generator=gr) * 3 - gallery[ref_i] * 0.5
    fiq_rows = [(names[int(r)], names[int(t)], [f"cap a {i}", f"cap b {i}"])
                for i, (r, t) in enumerate(zip(ref_i, tgt_i))]
    fm = FakeModel(text_f, D)
    r10, r50 = validate.compute_fiq_val_metrics(FakeFiqDataset(fiq_rows), fm, gallery, names,
                                                device=torch.device("cpu"))
    members = []
    for r, t in zip(ref_i, tgt_i):
        pool = [int(x) for x in torch.randperm(NG, generator=gr)[:12] if int(x) not in (int(r), int(t))][:5]
        m = pool + [int(t)]
        members.append([names[j] for j in m])
    cirr_rows = [(names[int(r)], names[int(t)], f"cap {i}", members[i])
                 for i, (r, t) in enumerate(zip(ref_i, tgt_i))]
    fm = FakeModel(text_f, D)
    cirr = validate.compute_cirr_val_metrics(FakeCirrDataset(cirr_rows), fm, gallery, names,
                                             device=torch.device("cpu"))
    pred = torch.nn.functional.normalize(gallery[ref_i] + text_f, dim=-1)
    gal_n = torch.nn.functional.normalize(gallery, dim=-1).float()
    order = torch.argsort(1 - pred @ gal_n.T, dim=-1)[:, :50]
    np.savez_compressed(os.path.join(OUT, "recall.npz"), gallery=gallery.numpy(), text_feats=text_f.numpy(),
                        ref_idx=ref_i.numpy(), tgt_idx=tgt_i.numpy(),
                        members=json.dumps(members), names=json.dumps(names),
                        fiq=np.array([r10, r50]), cirr=np.array(cirr), pred=pred.numpy(),
                        top50=order.numpy())

    # ---- 6. AdamW -------------------------------------------------------------------------
    ga = torch.Generator().manual_seed(11)
    p0 = torch.randn(1000, generator=ga)
    g1 = torch.randn(1000, generator=ga) * 0.1
    g2 = torch.randn(1000, generator=ga) * 0.1
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([{"params": [p], "lr": 2e-5, "betas": (0.9, 0.999), "eps": 1e-7}])
    p.grad = g1.clone(); opt.step(); p1 = p.detach().clone()
    p.grad = g2.clone(); opt.step(); p2 = p.detach().clone()
    np.savez_compressed(os.path.join(OUT, "adamw.npz"), p0=p0.numpy(), g1=g1.numpy(), g2=g2.numpy(),
                        p1=p1.numpy(), p2=p2.numpy(), lr=np.float64(2e-5))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
