"""bf16 noise floor of the full-size gradient tests (tests/test_fullsize_gpu.py).  Needs only this repository (the CPU
oracle); no /root/reference.      python tests/golden/make_noise_floor.py [case ...]   -> noise_floor.json

For every full-size case the oracle step runs twice on the CPU: in fp32, and with every matrix-product operand rounded to
bf16 with fp32 accumulation (oracle/noise.py - forward AND backward products, bias gradients summed from the rounded output
gradient; the arithmetic class of the MI355X kernels and, with fp16, of the reference's own autocast path).  The per-tensor
relative L2 distance between the two gradient sets is what operand rounding alone costs on THIS model, batch and seed.  It is a
random variable of the rounding pattern - gradients that are remainders of cancelling sums (bias / LayerNorm vectors at B = 4)
move by up to 1.7x between equally legitimate patterns - so it is sampled on four rounding grids (x -> bf16(x s) / s,
s = 2^(r/4)) and the floor of a tensor is the RMS over the samples (`operands`; the largest sample is kept as `operands_max`).
The GPU tests then assert
    HIP-vs-oracle error  <=  1.5 x floor     (per tensor; the ratio is printed)
instead of an absolute constant.  `autocast` = the same with every product's OUTPUT rounded too (the reference's storage
class under torch.cuda.amp.autocast) - recorded beside it for information.

Cases = exactly the seeded inputs of the tests:
  vitl14_b8            ViT-L/14 text step, B = 8, 40 000 x 768 bank                 (test_vitl14_every_gradient_matches_oracle)
  vitl14_b8_e4m3       the same on the dequantised e4m3 100 000-row bank            (test_fp8_bank_trainer_step_100k)
  config1_vitb32_b4    ViT-B/32 both towers, in-batch, B = 4                         (test_config1_vitb32_inbatch_step_every_gradient)
  blip_768 / blip_1024 BERT-base fusion 12 x 768 over 577 image tokens, B = 8        (test_blip_fusion_full_shape)
  blip_768_refinit / blip_1024_refinit   the same with BertPreTrainedModel's own init scale on every matrix (the rank-collapsing
                       regime the test avoids: its floor is on record to show WHY)
"""
import json
import os
import sys
import time

import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from cases import fusion_sd  # noqa: E402
from oracle import bank_loss, bert_fusion, clip_text, clip_vision, noise  # noqa: E402
from spn4cir_amd import synthetic  # noqa: E402


def _grads(step, sd):
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    out = step(params)
    out["loss"].backward()
    return out, {k: p.grad for k, p in params.items() if p.grad is not None and float(p.grad.abs().max()) > 0}


REALISATIONS = 4


def floors(step, sd, row_subset=None):
    """-> dict(loss_fp32, operands={tensor: RMS floor over REALISATIONS rounding grids}, operands_max={tensor: largest sample},
    autocast={tensor: floor with outputs rounded too, one sample}, feature / loss deviations of the first sample)"""
    torch.manual_seed(0)
    o32, g32 = _grads(step, sd)
    res = {"loss_fp32": float(o32["loss"].detach()), "realisations": REALISATIONS}

    def sample(outputs, r):
        with noise.bf16_gemm_operands(outputs=outputs, realisation=r) as m:
            ob, gb = _grads(step, sd)
        assert m.products > 0
        fl = {}
        for k, ref in g32.items():
            got = gb[k]
            if row_subset and k in row_subset:
                got, ref = got[row_subset[k]], ref[row_subset[k]]
            fl[k] = noise.rel_l2(got, ref)
        cos = torch.nn.functional.cosine_similarity(ob["feats"].double(), o32["feats"].double(), dim=-1)
        return fl, abs(float(ob["loss"].detach()) - float(o32["loss"].detach())), float((1 - cos).max())

    samples = []
    for r in range(REALISATIONS):
        fl, dl, dc = sample(False, r)
        samples.append(fl)
        if r == 0:
            res["operands_loss_abs_diff"], res["operands_feat_max_1_minus_cos"] = dl, dc
    res["operands"] = {k: (sum(s_[k] ** 2 for s_ in samples) / len(samples)) ** 0.5 for k in samples[0]}
    res["operands_max"] = {k: max(s_[k] for s_ in samples) for k in samples[0]}
    res["autocast"], res["autocast_loss_abs_diff"], res["autocast_feat_max_1_minus_cos"] = sample(True, 0)
    return res


def case_vitl14(e4m3=False):
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    B, M, tau = 8, (100000 if e4m3 else 40000), 0.02
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    ids = synthetic.token_ids(B, seed=1)
    target, refer = synthetic.banks(M, D, seed=2)
    ridx, labels = synthetic.triplet_indices(B, M, seed=4)
    if e4m3:
        data, scale = bank_loss.quantize_e4m3(target)
        target = bank_loss.dequantize_e4m3(data, scale)[:, :D]

    def step(params):
        feats = clip_text.encode_text(params, ids.long())
        return {"feats": feats.detach(), "loss": bank_loss.bank_large_step(refer, ridx, feats, target, labels, tau)}
    return floors(step, sd)


def case_config1():
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-B/32"]
    B, tau = 4, 0.01
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    sd.update(clip_vision.synthetic_vision_state_dict(768, 12, 32, 224, D, seed=5))
    ids = synthetic.token_ids(B, seed=1)
    g = torch.Generator().manual_seed(0)
    ref_img, tgt_img = torch.randn(B, 3, 224, 224, generator=g), torch.randn(B, 3, 224, 224, generator=g)

    def step(params):
        t = clip_text.encode_text(params, ids.long())
        loss = bank_loss.inbatch_step(clip_vision.encode_image(params, ref_img), t, clip_vision.encode_image(params, tgt_img), tau)
        return {"feats": t.detach(), "loss": loss}
    rows = {"token_embedding.weight": torch.unique(ids.long())}
    return floors(step, sd, rows)


def case_blip(enc_width, init):
    W, layers, heads, I, Dp, vocab, max_pos = 768, 12, 12, 3072, 256, 30524, 512
    B, L, S, M, tau, b = 128, 32, 577, 30000, 0.03, 8
    sd = fusion_sd(layers, W, I, enc_width, Dp, vocab, max_pos, seed=0, init=init)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 30522, (B, L), generator=g, dtype=torch.int32)
    ids[:, 0] = 30523
    lens = torch.randint(6, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int32)
    ids = ids * mask
    enc = torch.randn(B, S, enc_width, generator=g)
    bank = torch.nn.functional.normalize(torch.randn(M, Dp, generator=g))
    labels = torch.randint(0, M, (B,), generator=g)

    def step(params):
        q = bert_fusion.fusion_query(params, ids[:b], mask[:b], enc[:b])
        return {"feats": q.detach(), "loss": torch.nn.functional.cross_entropy(q @ bank.t() / tau, labels[:b])}
    rows = {"embeddings.position_embeddings.weight": slice(0, L)}
    return floors(step, sd, rows)


CASES = {
    "vitl14_b8": lambda: case_vitl14(False),
    "vitl14_b8_e4m3": lambda: case_vitl14(True),
    "config1_vitb32_b4": case_config1,
    "blip_768": lambda: case_blip(768, "small_residual"),
    "blip_1024": lambda: case_blip(1024, "small_residual"),
    "blip_768_refinit": lambda: case_blip(768, "reference"),
    "blip_1024_refinit": lambda: case_blip(1024, "reference"),
}


def main():
    path = os.path.join(OUT, "noise_floor.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for name in (sys.argv[1:] or list(CASES)):
        t0 = time.time()
        r = CASES[name]()
        for tag in ("operands", "operands_max", "autocast"):
            r[tag] = {k: float(f"{v:.4e}") for k, v in r[tag].items()}
        data[name] = r
        ops_ = r["operands"]
        wk = max(ops_, key=ops_.get)
        print(f"{name}: {len(ops_)} tensors, worst operand floor {ops_[wk]:.3e} ({wk}), median {sorted(ops_.values())[len(ops_) // 2]:.3e}, "
              f"feature 1-cos {r['operands_feat_max_1_minus_cos']:.2e}, |dloss| {r['operands_loss_abs_diff']:.2e}  [{time.time() - t0:.0f} s]",
              flush=True)
        with open(path, "w") as f:
            json.dump(data, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
