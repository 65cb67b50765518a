"""The reference's BLIP loop body (blip4cir/train.py:110-129) on the drop-in blip_models.CIRPlus: caption STRINGS (host WordPiece
every step, padding='longest'), reference tokens gathered from the token bank by image id, autograd backward into the exposed
parameters + the learnable temperature, AdamW, GradScaler as in the reference - beside fusion.BlipStage2Trainer on the same shapes
(B = 128, captions of 5..30 words, 577 x 768 image tokens, 30 000 x 256 target bank).  Config 4's shape on one GPU.
    python tools/dropin_blip_step.py [--fused-optim] [--images 4000]"""
import os, random, sys, tempfile, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd.blip_models import CIRPlus

WORDS = ("is red and has long sleeves with a floral pattern shorter more colorful dress shirt top blue striped darker "
         "lighter green collar buttons casual formal the same but different style material lace silk cotton").split()


def synthetic_state_dict(g, W=768, layers=12, I=3072, E=768, Dp=256, vocab=30524, max_pos=512):
    sd = {"text_encoder.embeddings.word_embeddings.weight": torch.randn(vocab, W, generator=g) * 0.02,
          "text_encoder.embeddings.position_embeddings.weight": torch.randn(max_pos, W, generator=g) * 0.02,
          "text_encoder.embeddings.LayerNorm.weight": torch.ones(W), "text_encoder.embeddings.LayerNorm.bias": torch.zeros(W),
          "text_proj.weight": torch.randn(Dp, W, generator=g) * 0.02, "text_proj.bias": torch.zeros(Dp)}
    lin = lambda o, i: torch.randn(o, i, generator=g) * 0.02
    for l in range(layers):
        p = f"text_encoder.encoder.layer.{l}."
        for att, kv in (("attention", W), ("crossattention", E)):
            for n, i in (("query", W), ("key", kv), ("value", kv)):
                sd[p + f"{att}.self.{n}.weight"], sd[p + f"{att}.self.{n}.bias"] = lin(W, i), torch.zeros(W)
            sd[p + f"{att}.output.dense.weight"], sd[p + f"{att}.output.dense.bias"] = lin(W, W), torch.zeros(W)
            sd[p + f"{att}.output.LayerNorm.weight"], sd[p + f"{att}.output.LayerNorm.bias"] = torch.ones(W), torch.zeros(W)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = lin(I, W), torch.zeros(I)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = lin(W, I), torch.zeros(W)
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = torch.ones(W), torch.zeros(W)
    return sd


def run(fused_optim=False, images=4000, steps=10, warmup=3):
    B, M, S, E = 128, 30000, 577, 768
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    # a bert-base-uncased shaped vocabulary: specials, filler entries, the caption words (the real file is not available offline)
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS
    vocab += [f"tok{i}" for i in range(30522 - len(vocab))]
    vf = os.path.join(tempfile.mkdtemp(), "vocab.txt")
    with open(vf, "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    model = CIRPlus(synthetic_state_dict(g), tau=0.03, device=dev, plus=True, vocab_file=vf)
    bank = torch.empty(images, S, E, dtype=torch.bfloat16, device=dev)
    dg = torch.Generator(device=dev).manual_seed(5)
    for s0 in range(0, images, 500):
        n = min(500, images - s0)
        bank[s0:s0 + n].copy_(torch.randn(n, S, E, generator=dg, device=dev))
    model.refer_bank = bank                             # already the device image (ops.token_bank_bf16 keeps it)
    model.target_bank = torch.nn.functional.normalize(torch.randn(M, 256, generator=g))
    random.seed(0)
    caps = [" ".join(random.choice(WORDS) for _ in range(random.randint(5, 30))) for _ in range(B)]
    refer_ids = torch.randint(0, images, (B,), generator=g)
    target_ids = torch.randint(0, M, (B,), generator=g)
    indexs = torch.arange(B)
    params = [p for p in model.parameters() if p.requires_grad]
    if fused_optim:
        from spn4cir_amd.optim import AdamW
        opt = AdamW(params, lr=5e-6, betas=(0.9, 0.999), eps=1e-7)
    else:
        opt = torch.optim.AdamW(params, lr=5e-6, betas=(0.9, 0.999), eps=1e-7)
    scaler = torch.cuda.amp.GradScaler()
    model.blip.eval()                                   # train.py:111

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.forward(caps, indexs, target_ids, refer_ids)["bank_loss"]
        "{:05.3f}".format(loss)                         # train.py:118-120 formats the loss every step: a host sync, kept
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        model.parameters_changed()
        return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t, n = time.perf_counter(), steps
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    ids, mask = model.tokenize(caps)
    return {"optimizer": "spn4cir_amd.optim.AdamW" if fused_optim else "torch.optim.AdamW", "ms_per_step": round(dt * 1e3, 3),
            "triplets_per_s": round(B / dt, 1), "loss_last": round(float(loss.item()), 5),
            "text_rows_live": f"{int(mask.sum())} of {mask.numel()}"}


if __name__ == "__main__":
    r = run("--fused-optim" in sys.argv, int(sys.argv[sys.argv.index("--images") + 1]) if "--images" in sys.argv else 4000)
    print(f"blip drop-in loop ({r['optimizer']}): {r['ms_per_step']:.2f} ms/step  {r['triplets_per_s']:.0f} triplets/s  "
          f"loss {r['loss_last']:.4f}  text rows live {r['text_rows_live']}")
