"""Pin the CPU oracle against vectors captured from the reference itself (tests/golden/)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import bank_loss, clip_text, clip_vision, optim, recall
from cases import LOSS_CASES, loss_case_inputs


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _tiny_sd(golden_dir):
    z = _load(golden_dir, "tiny_clip.npz")
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return z, sd


def test_text_tower_matches_reference(golden_dir):
    z, sd = _tiny_sd(golden_dir)
    ids = torch.from_numpy(z["ids"])
    feats, hidden = clip_text.encode_text(sd, ids, return_hidden=True)
    assert torch.allclose(feats, torch.from_numpy(z["text_feats"]), atol=2e-6, rtol=1e-5)
    for i in range(2):
        ref = torch.from_numpy(z[f"hidden_{i}"])
        # rows after the EOT token see only -inf-free causal context too, so all rows compare
        assert torch.allclose(hidden[i + 1], ref, atol=5e-6, rtol=1e-5), i


def test_vision_tower_matches_reference(golden_dir):
    z, sd = _tiny_sd(golden_dir)
    out = clip_vision.encode_image(sd, torch.from_numpy(z["image"]))
    assert torch.allclose(out, torch.from_numpy(z["image_feats"]), atol=2e-6, rtol=1e-5)


def test_cirplus_step_loss_and_grads(golden_dir):
    z, sd = _tiny_sd(golden_dir)
    s = _load(golden_dir, "cirplus_step.npz")
    ids = torch.from_numpy(z["ids"])
    text_keys = [k for k in sd if not k.startswith("visual.") and k != "logit_scale"]
    params = {k: sd[k].clone().requires_grad_(True) for k in text_keys}
    feats = clip_text.encode_text(params, ids)
    loss = bank_loss.bank_large_step(torch.from_numpy(s["refer_bank"]), torch.from_numpy(s["ref_img_ids"]),
                                     feats, torch.from_numpy(s["target_bank"]),
                                     torch.from_numpy(s["tgt_img_ids"]), float(s["tau"]))
    assert abs(loss.item() - float(s["loss_plus"])) < 2e-5
    loss.backward()
    checked = 0
    for k in text_keys:
        ref = torch.from_numpy(s["grad_plus::" + k])
        got = params[k].grad
        denom = ref.abs().max().clamp_min(1e-8)
        assert (got - ref).abs().max() / denom < 2e-4, k
        checked += 1
    assert checked == len(text_keys) and checked == 29
    # per-triplet reference row (plus=False), zscir-style
    feats2 = clip_text.encode_text(sd, ids)
    loss2 = bank_loss.bank_large_step(torch.from_numpy(s["trip_bank"]), torch.arange(ids.shape[0]), feats2,
                                      torch.from_numpy(s["target_bank"]), torch.from_numpy(s["tgt_img_ids"]),
                                      float(s["tau"]))
    assert abs(loss2.item() - float(s["loss_trip"])) < 2e-5


@pytest.mark.parametrize("ci", range(len(LOSS_CASES)))
def test_bank_loss_cases(golden_dir, ci):
    z = _load(golden_dir, "loss_cases.npz")
    text, rb, bank, ridx, labels, tau = loss_case_inputs(ci)
    text.requires_grad_(True)
    loss = bank_loss.bank_large_step(rb, ridx, text, bank, labels, tau)
    loss.backward()
    assert abs(loss.item() - float(z[f"c{ci}_loss"])) < 1e-4 * max(1.0, abs(float(z[f"c{ci}_loss"])))
    ref = torch.from_numpy(z[f"c{ci}_dtext"])
    assert (text.grad - ref).abs().max() / ref.abs().max() < 1e-4
    # the closed-form fp64 pieces used by the kernel tests agree with autograd
    q = bank_loss.l2_normalize(rb[ridx] + text.detach())
    lse, lab, row = bank_loss.infonce_stats(q, bank, labels, tau)
    assert abs(row.mean().item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))


def test_recall_matches_reference(golden_dir):
    z = _load(golden_dir, "recall.npz")
    names = json.loads(str(z["names"]))
    members = json.loads(str(z["members"]))
    gallery = z["gallery"]
    ref_idx, tgt_idx = z["ref_idx"], z["tgt_idx"]
    pred = bank_loss.l2_normalize(torch.from_numpy(gallery[ref_idx] + z["text_feats"])).numpy()
    assert np.allclose(pred, z["pred"], atol=1e-6)
    ref_names = [names[i] for i in ref_idx]
    tgt_names = [names[i] for i in tgt_idx]
    r10, r50 = recall.fiq_recall(pred, gallery, names, tgt_names, ref_names)
    assert (r10, r50) == pytest.approx(tuple(z["fiq"]), abs=1e-9)
    cirr = recall.cirr_recall(pred, gallery, names, ref_names, tgt_names, members)
    assert cirr == pytest.approx(tuple(z["cirr"]), abs=1e-4)
    order, _ = recall.ranked_indices(pred, gallery)
    same = [set(order[i, :50]) == set(z["top50"][i]) for i in range(len(order))]
    assert all(same)          # identical top-50 index sets
    assert 5.0 < r10 < 95.0   # the synthetic task is neither trivial nor impossible


def test_adamw_matches_torch(golden_dir):
    z = _load(golden_dir, "adamw.npz")
    p = torch.from_numpy(z["p0"]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    optim.adamw_step(p, torch.from_numpy(z["g1"]), m, v, 1, float(z["lr"]))
    assert torch.allclose(p, torch.from_numpy(z["p1"]), atol=1e-7, rtol=1e-6)
    optim.adamw_step(p, torch.from_numpy(z["g2"]), m, v, 2, float(z["lr"]))
    assert torch.allclose(p, torch.from_numpy(z["p2"]), atol=1e-7, rtol=1e-6)


def test_blip_fusion_matches_reference(golden_dir):
    from oracle import bert_fusion
    z = _load(golden_dir, "blip_fusion.npz")
    sd = {k[4:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("sd::")}
    ids, mask, enc = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"]), torch.from_numpy(z["enc"])
    h = bert_fusion.fusion_forward(sd, ids, mask, enc)
    valid = mask.bool()
    assert torch.allclose(h[valid], torch.from_numpy(z["last_hidden_state"])[valid], atol=2e-5, rtol=1e-4)
    q = bert_fusion.fusion_query(sd, ids, mask, enc)
    assert torch.allclose(q, torch.from_numpy(z["q"]), atol=1e-5)
    loss = torch.nn.functional.cross_entropy((q @ torch.from_numpy(z["bank"]).T) / float(z["tau"]),
                                             torch.from_numpy(z["labels"]))
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    n = 0
    for k in sd:
        if "grad::" + k in z.files:
            ref = torch.from_numpy(z["grad::" + k])
            assert (sd[k].grad - ref).abs().max() <= 2e-4 * ref.abs().max().clamp_min(1e-6) + 1e-7, k
            n += 1
    assert n >= 50


def test_inbatch_step_loss_and_grads(golden_dir):
    """BASELINE config 1 (clip4cir/models.py:151-167, wo_bank=True): the oracle's in-batch step reproduces
    the reference's bbc_loss and the gradient of every parameter, visual tower included."""
    from oracle import bank_loss, clip_text, clip_vision
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    s = np.load(os.path.join(golden_dir, "cirplus_inbatch.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("sd::")
          and z[k].dtype == np.float32}
    ref = clip_vision.encode_image(sd, torch.from_numpy(s["refer_image"]))
    tgt = clip_vision.encode_image(sd, torch.from_numpy(s["target_image"]))
    assert torch.allclose(ref.detach(), torch.from_numpy(s["refer_feats"]), atol=1e-5)
    text = clip_text.encode_text(sd, torch.from_numpy(s["ids"]))
    loss = bank_loss.inbatch_step(ref, text, tgt, float(s["tau"]))
    assert abs(loss.item() - float(s["loss"])) < 1e-5
    loss.backward()
    n = 0
    for k in s.files:
        if not k.startswith("grad::"):
            continue
        g, r = sd[k[6:]].grad, torch.from_numpy(s[k])
        assert g is not None, k
        assert (g - r).norm() <= 1e-4 * r.norm() + 1e-7, (k, (g - r).norm().item(), r.norm().item())
        n += 1
    assert n == 61


def test_e4m3_quantizer_known_answers():
    """OCP e4m3fn known values (bias 7, 3 mantissa bits, max 448, RNE): the oracle's fp8 bank format."""
    from oracle import bank_loss
    row = torch.tensor([[448.0, 1.0, 1.0625, 1.1875, -0.015625, 2.0 ** -9, 0.0, 240.0]])
    data, scale = bank_loss.quantize_e4m3(row)
    assert scale.item() == 1.0
    # 448 = 0x7E; 1.0 = 0x38; 1.0625 ties to even -> 1.0; 1.1875 ties to even -> 1.25 = 0x3A;
    # -2^-6 = 0x88 (smallest normal, negative); 2^-9 = smallest subnormal 0x01; 0; 240 = 0x77
    assert data[0].tolist() == [0x7E, 0x38, 0x38, 0x3A, 0x88, 0x01, 0x00, 0x77]
    g = torch.Generator().manual_seed(0)
    bank = torch.nn.functional.normalize(torch.randn(64, 96, generator=g))
    data, scale = bank_loss.quantize_e4m3(bank)
    deq = bank_loss.dequantize_e4m3(data, scale)
    assert (deq - bank).abs().max() <= bank.abs().amax() * 2.0 ** -4          # half an ulp of 3 mantissa bits
    assert torch.equal(bank_loss.quantize_e4m3(deq)[0], data)                  # idempotent


def test_resnet_tower_matches_reference(golden_dir):
    """ModifiedResNet restatement (oracle/clip_resnet.py) against the reference's encode_image on a tiny RN CLIP."""
    from oracle import clip_resnet
    z = np.load(os.path.join(golden_dir, "tiny_clip_resnet.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    cfg = clip_resnet.resnet_cfg_from_state_dict(sd)
    assert cfg["layers"] == (1, 2, 1, 1) and cfg["width"] == 8 and cfg["res"] == 64 and cfg["embed_dim"] == 128
    out = clip_resnet.encode_image(sd, torch.from_numpy(z["image"]))
    ref = torch.from_numpy(z["image_feats"])
    assert (out - ref).abs().max() < 1e-5 * ref.abs().max().clamp_min(1.0)


def test_tokmax_oracle_loop_equals_batched_form():
    """oracle.bank_loss.tokmax_infonce (per-sample loop, as the reference writes it) == einsum + amax + CE."""
    from oracle import bank_loss
    g = torch.Generator().manual_seed(0)
    q = torch.nn.functional.normalize(torch.randn(6, 32, generator=g), dim=-1).double()
    bank = torch.nn.functional.normalize(torch.randn(50, 32, 32, generator=g), dim=-1).double()
    labels = torch.randint(0, 50, (6,), generator=g)
    a = bank_loss.tokmax_infonce(q, bank, labels, 0.07)
    b = torch.nn.functional.cross_entropy(torch.einsum("bd,mkd->bmk", q, bank).amax(-1) / 0.07, labels)
    assert abs(a.item() - b.item()) < 1e-12


def test_tgcir_oracle_matches_reference(golden_dir):
    """oracle.tgcir_head == tgcir/models.py CIRPlus.forward on CPU: loss, query, mod tokens, every gradient."""
    import os
    from oracle import tgcir_head
    from cases import TGCIR, tgcir_grad_check, tgcir_inputs, tgcir_weights
    sd, head = tgcir_weights()
    ids, ref, bank, labels = tgcir_inputs()
    z = np.load(os.path.join(golden_dir, "tgcir_step.npz"))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    head = {k: v.clone().requires_grad_(True) for k, v in head.items()}
    loss, q = tgcir_head.bank_step(sd, head, ids, ref, bank, labels, TGCIR["TAU"])
    tokens, feats = tgcir_head.text_tokens(sd, ids)
    mod = tgcir_head.extract_text_fea(tokens, feats, head)
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    assert (q.detach() - torch.from_numpy(z["q"])).abs().max() < 1e-5
    assert (mod.detach() - torch.from_numpy(z["mod_token"])).abs().max() < 1e-5
    loss.backward()
    for k in tgcir_head.HEAD_KEYS:
        tgcir_grad_check(z, k, head[k].grad, 2e-4)
    for k, v in sd.items():
        tgcir_grad_check(z, "clip." + k, v.grad, 2e-4)


def test_tgcir_oracle_image_side_matches_reference(golden_dir):
    """oracle.tgcir_head.img_embed == tgcir CIRPlus.img_embed(return_pool_and_normalized=True) on CPU."""
    import os
    from oracle import tgcir_head
    from cases import tgcir_image_side
    vsd, ihead, images = tgcir_image_side()
    z = np.load(os.path.join(golden_dir, "tgcir_step.npz"))
    emb, pooled = tgcir_head.img_embed(vsd, ihead, images)
    assert (emb - torch.from_numpy(z["img_tokens"])).abs().max() < 2e-5
    assert (pooled - torch.from_numpy(z["img_pooled"])).abs().max() < 2e-6


def test_query_split_e4m3_model():
    """oracle.bank_loss.split_query_e4m3 (the model of the fp8-MFMA bank pass): two e4m3 terms carry a query to ~2^-8 of its
    largest element, a zero row stays zero, and the split of an exactly representable row is exact."""
    from oracle import bank_loss
    g = torch.Generator().manual_seed(11)
    q = torch.nn.functional.normalize(torch.randn(64, 768, generator=g), dim=1).bfloat16().float()
    m = bank_loss.split_query_e4m3(q)
    err = (m - q).abs().amax(dim=1) / q.abs().amax(dim=1)
    assert err.max().item() < 2.0 ** -7
    one = bank_loss.split_query_e4m3(q[:1])                 # per-row: independent of the other rows
    assert torch.equal(one, m[:1])
    z = torch.zeros(2, 768)
    assert torch.equal(bank_loss.split_query_e4m3(z), z)
    exact = torch.tensor([[448.0, -224.0, 1.0, 0.5, 0.0, 28.0, -0.015625, 2.0]])      # e4m3 values at scale 1
    assert torch.equal(bank_loss.split_query_e4m3(exact), exact)


def test_blip_vit_oracle_matches_reference(golden_dir):
    """oracle.blip_vit vs vectors captured from blip4cir/vit.py's own Block / Attention / VisionTransformer.forward
    (tests/golden/make_golden_blipvit.py): block outputs, the attention branch alone, the tower's token sequence and the
    pooled feature."""
    import os
    from oracle import blip_vit
    z = np.load(os.path.join(golden_dir, "blip_vit.npz"))
    for tag in ("blkA", "blkB"):
        sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(tag + ".sd.")}
        x, heads = torch.from_numpy(z[tag + ".x"]), int(z[tag + ".heads"])
        W = x.shape[-1]
        y = blip_vit.block(sd, "", x, heads)
        assert (y - torch.from_numpy(z[tag + ".y"])).abs().max() < 2e-5, tag
        h = torch.nn.functional.layer_norm(x, (W,), sd["norm1.weight"], sd["norm1.bias"], 1e-6)
        a = blip_vit.attention(sd, "", h, heads)
        assert (a - torch.from_numpy(z[tag + ".attn_out"])).abs().max() < 2e-5, tag
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    tokens, pooled = blip_vit.img_embed(sd, torch.from_numpy(z["image"]), int(z["heads"]))
    assert (tokens - torch.from_numpy(z["tokens"])).abs().max() < 2e-5
    assert (pooled - torch.from_numpy(z["pooled"])).abs().max() < 2e-6


def test_tokmax_oracle_matches_reference(golden_dir):
    """oracle.bank_loss.tokmax_infonce vs the reference's forward_stage2 (blip2_qformer_cir_align_prompt.py:247-268,
    captured by tests/golden/make_golden_blip2.py): loss, d loss / d fusion_feats, d loss / d temp - including the case
    with tied token rows inside a target."""
    import os
    from cases import BLIP2_CASES, blip2_target_feats
    from oracle import bank_loss
    z = np.load(os.path.join(golden_dir, "blip2_stage2.npz"))
    for tag in BLIP2_CASES:
        feats = torch.from_numpy(z[f"{tag}.fusion_feats"]).clone().requires_grad_(True)
        temp = torch.tensor(float(z[f"{tag}.temp"]), requires_grad=True)
        loss = bank_loss.tokmax_infonce(feats, blip2_target_feats(tag), torch.from_numpy(z[f"{tag}.target_indexs"]), temp)
        loss.backward()
        assert abs(loss.item() - float(z[f"{tag}.loss_qtc"])) < 2e-5 * max(1.0, abs(loss.item())), tag
        ref = torch.from_numpy(z[f"{tag}.d_fusion_feats"])
        assert (feats.grad - ref).abs().max() < 1e-5 * max(1.0, ref.abs().max().item()), tag
        assert abs(temp.grad.item() - float(z[f"{tag}.d_temp"])) < 1e-4 * max(1.0, abs(float(z[f"{tag}.d_temp"]))), tag


def test_negtype_oracle_matches_reference_capture(golden_dir):
    """oracle/negtype.py against the reference's own text_neg_loss / refer_neg_loss / infonce_loss + forward()'s neg_type mask
    (clip4cir/models_negtype.py:53-134; tests/golden/make_golden_negtype.py): loss and the three feature gradients."""
    import os
    import numpy as np
    import torch
    from oracle import negtype
    z = np.load(os.path.join(golden_dir, "negtype.npz"))
    for tag in ("a", "b"):
        r, t, i = (torch.from_numpy(z[f"{tag}::{k}"]) for k in ("refer", "text", "target"))
        tau = float(z[f"{tag}::tau"])
        for nt in (1, 2, 4, 8, 7, 15, 5, 10):
            rr, tt, ii = (x.clone().requires_grad_(True) for x in (r, t, i))
            loss = negtype.loss(rr, tt, ii, tau, nt)
            loss.backward()
            assert abs(loss.item() - float(z[f"{tag}::{nt}::loss"])) < 2e-6 * max(1.0, abs(loss.item()))
            for g, k in ((rr, "refer"), (tt, "text"), (ii, "target")):
                ref = torch.from_numpy(z[f"{tag}::{nt}::d_{k}"])
                assert (g.grad - ref).abs().max() < 1e-6 * max(1.0, ref.abs().max().item()), (tag, nt, k)


def test_blip_val_chain_oracle_matches_reference(golden_dir):
    """blip_val.npz = the reference's own blip4cir chain on caption STRINGS (utils.extract_index_features ->
    BLIP_Retrieval.img_embed incl. blip_cir.py:62, validate.generate_*_val_predictions -> img_txt_fusion incl. the tokenizer
    call of blip_cir.py:87-88, models.CIRPlus.forward + backward).  The oracle pieces (blip_vit.img_embed, the WordPiece
    restatement, bert_fusion.fusion_query, cross-entropy) must reproduce it in fp32."""
    from cases import BLIPVAL, blipval_inputs, blipval_state_dict, blipval_vocab
    from oracle import bert_fusion, blip_vit
    from spn4cir_amd.bert_tokenizer import BertWordPieceTokenizer
    z = _load(golden_dir, "blip_val.npz")
    c, sd, inp = BLIPVAL, blipval_state_dict(), blipval_inputs()
    tok = BertWordPieceTokenizer(vocab=blipval_vocab())
    with torch.no_grad():
        tokens, pooled = blip_vit.img_embed(sd, inp["images"], c["W"] // 64)
    assert torch.allclose(pooled, torch.from_numpy(z["index_features_p"]), atol=2e-5)           # blip_cir.py:62 pinned
    assert torch.allclose(tokens[:, ::48, ::16], torch.from_numpy(z["index_tokens_sample"]), atol=1e-4, rtol=1e-4)
    assert torch.allclose(tokens.norm(dim=-1), torch.from_numpy(z["index_tokens_norm"]), rtol=1e-4)
    fsd = {k[len("text_encoder."):]: v for k, v in sd.items() if k.startswith("text_encoder.")}
    fsd.update({k: v for k, v in sd.items() if k.startswith("text_proj.")})
    name2i = {n: i for i, n in enumerate(inp["names"])}

    def fuse(rows_ref, caps, params=fsd):
        ids, mask = tok.enc_batch(caps)
        ref = tokens[torch.tensor([name2i[r] for r in rows_ref])]
        return bert_fusion.fusion_query(params, ids.to(torch.int32), mask.to(torch.int32), ref)

    caps = [f"{r[2][0].strip('.?, ').capitalize()} and {r[2][1].strip('.?, ')}" for r in inp["fiq_rows"]]
    with torch.no_grad():
        pf = fuse([r[0] for r in inp["fiq_rows"]], caps)
        pc = fuse([r[0] for r in inp["cirr_rows"]], [r[2] for r in inp["cirr_rows"]])
    assert torch.allclose(pf, torch.from_numpy(z["pred_fiq"]), atol=2e-5)
    assert torch.allclose(pc, torch.from_numpy(z["pred_cirr"]), atol=2e-5)
    # training step on strings
    caps, _, target_ids, refer_ids = inp["train"]
    params = {k: v.clone().requires_grad_(True) for k, v in fsd.items()}
    tau = torch.tensor(c["TAU"], requires_grad=True)
    ids, mask = tok.enc_batch(caps)
    assert ids.shape[1] == int(z["train_ids_longest"])
    q = bert_fusion.fusion_query(params, ids.to(torch.int32), mask.to(torch.int32), tokens[refer_ids])
    loss = torch.nn.functional.cross_entropy((q @ pooled.T) / tau, target_ids)
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    assert abs(tau.grad.item() - float(z["dtau"])) < 1e-3 * abs(float(z["dtau"]))
    n = 0
    for k in z.files:
        if k.startswith("grad::"):
            name = k[6:]
            name = name[len("text_encoder."):] if name.startswith("text_encoder.") else name
            ref = torch.from_numpy(z[k])
            assert (params[name].grad - ref).abs().max() <= 5e-4 * ref.abs().max().clamp_min(1e-6) + 1e-7, name
            n += 1
    assert n >= 10
