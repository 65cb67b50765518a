"""Parity at BASELINE.json's full model sizes (round-2 VERDICT item 1): the configurations the small golden fixtures do
not reach - ViT-L/14's text tower with every parameter gradient against the oracle's autograd, the packed (dead-token-free)
mode against the dense one at B = 256, the BLIP fusion encoder at 12 layers x 768 x 577 image tokens (enc_width 768 and
1024), and a trainer step over a 100 000-row fp8 bank.

Gates (same as the small-fixture tests): features 1 - cos <= 1e-3 (north_star), loss within 1e-2, per-parameter gradient
relative L2 against the fp32 CPU oracle (bf16 operands, fp32 accumulation) gated at 2 x the observed worst of each test (printed)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAD_GATE_VITL14 = 2.5e-2    # per-parameter relative L2 of the ViT-L/14 text-tower gradients: 2 x the observed worst (1.17e-2 on the
                             # bf16 bank in all three grouping modes, 1.23e-2 on the e4m3 bank; both on a LayerNorm weight of block 0)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _oracle_text_step(sd, ids, refer, ridx, target, labels, tau):
    from oracle import bank_loss, clip_text
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feats = clip_text.encode_text(params, ids.long())
    loss = bank_loss.bank_large_step(refer, ridx, feats, target, labels, tau)
    loss.backward()
    return feats.detach(), loss.item(), {k: v.grad for k, v in params.items()}


@pytest.mark.parametrize("groups", [None, [12], [7, 3, 2]], ids=["library-default", "one-group", "groups-7-3-2"])
def test_vitl14_every_gradient_matches_oracle(groups):
    """ViT-L/14 text tower (12 x 768, 124 M parameters), B = 8, M = 40 000: the loss and EVERY parameter gradient of the
    step (clip4cir/models_negplus.py:130-154 + autograd) against the oracle, through the library's own deferred
    weight-gradient path (spn_text_bwd), one explicit 12-block group and the data-parallel 7 + 3 + 2 grouping."""
    _need_gpu()
    from spn4cir_amd import ops, synthetic
    from spn4cir_amd.text_tower import TextTower
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    B, M, tau = 8, 40000, 0.02
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    ids = synthetic.token_ids(B, seed=1)
    target, refer = synthetic.banks(M, D, seed=2)
    ridx, labels = synthetic.triplet_indices(B, M, seed=4)
    t = TextTower(W, layers, heads, D, device="cuda")
    t.load_clip_state_dict(sd)
    feats = t.forward(ids.cuda())
    q, qb, inv = ops.combine_l2norm_fwd(refer.cuda(), ridx.cuda(), feats)
    bank_b = ops.prepare_bank(target.cuda())
    stats = ops.bank_stats_fwd(qb, bank_b, labels.cuda(), 1.0 / tau)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    dq = ops.bank_grad_q(qb, bank_b, labels.cuda(), 1.0 / tau, lse, 1.0 / B)[:, :D].contiguous()
    dtext = ops.combine_l2norm_bwd(q, inv, dq)
    if groups is None:
        grads = t.backward(dtext)
    else:
        grads = t.backward_phased(dtext, lambda a, b: None, groups)
    torch.cuda.synchronize()
    f_ref, loss_ref, g_ref = _oracle_text_step(sd, ids, refer, ridx, target, labels, tau)
    cos = torch.nn.functional.cosine_similarity(feats.cpu().double(), f_ref.double(), dim=-1)
    assert (1 - cos).max() < 1e-3
    assert abs(mean.item() - loss_ref) < 1e-2 * max(1.0, abs(loss_ref))
    views = t.named_views(grads)
    worst = {}
    for k, ref in g_ref.items():
        assert ref is not None and ref.norm() > 0, k
        worst[k] = _rel(views[k].cpu(), ref)
    wk = max(worst, key=worst.get)
    print(f"ViT-L/14 text step (groups {groups}): worst gradient error {worst[wk]:.3e} ({wk})")
    bad = {k: v for k, v in worst.items() if not v < GRAD_GATE_VITL14}
    assert not bad, bad
    assert len(worst) == len(sd) == 2 + 12 * layers + 3


def test_packed_matches_dense_at_full_size():
    """BASELINE config 2 (B = 256, ViT-L/14, 77-token rows of which ~27 % are live): the packed mode computes the same
    features (bit for bit per row: same kernels, same k order) and the same gradients up to the bf16 rounding of
    differently grouped sums (rows after EOT are dead under the causal mask, clip/model.py:330-336,356)."""
    _need_gpu()
    from spn4cir_amd import synthetic
    from spn4cir_amd.text_tower import TextTower
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    B = 256
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    ids = synthetic.token_ids(B, seed=1)
    t = TextTower(W, layers, heads, D, device="cuda")
    t.load_clip_state_dict(sd)
    dfeats = torch.randn(B, D, generator=torch.Generator().manual_seed(9)).cuda() / B
    f_dense = t.forward(ids.cuda()).clone()
    g_dense = t.backward(dfeats).clone()
    cu, total = TextTower.cu_seqlens(ids)
    assert total < B * 77 // 2
    f_packed = t.forward(ids.cuda(), cu.cuda(), total).clone()
    g_packed = t.backward(dfeats)
    assert (f_packed - f_dense).abs().max().item() <= 1e-6 * f_dense.abs().max().item()
    vd, vp = t.named_views(g_dense), t.named_views(g_packed)
    bad = {k: _rel(vp[k], vd[k]) for k in vd if vd[k].norm() > 0 and not _rel(vp[k], vd[k]) < 2e-2}
    assert not bad, bad


def _fusion_sd(layers, W, I, E, Dp, vocab, max_pos, seed):
    """BertModel(add_cross_attention) + text_proj state-dict with med.py's key names: BertPreTrainedModel's normal init scaled
    up on the query / key / value / intermediate matrices so that attention is not uniform, non-trivial LayerNorm affine, and
    SMALL residual-branch outputs (attention / cross-attention / FFN output.dense at std 0.01).  The last point keeps the
    12-layer post-LN stack from rank-collapsing: with all matrices at std 0.04 the mean pairwise cosine between the positions'
    hidden states grows 0.20, 0.45, 0.67, ... 0.9996, 0.9998 over the layers (every position carries the same vector at the
    top: the near-uniform cross-attention over 577 random image tokens adds one common vector per layer), the query / key
    gradients of the top layers are then the remainder of a cancelling sum and no bf16 attention backward reproduces them to
    better than ~0.2; with this init it ends at 0.47 and every tensor is held to the same gate."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
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


def test_fusion_init_is_not_rank_collapsed():
    """The claim the gradient gates of test_blip_fusion_full_shape rest on, checked on the oracle (CPU, 12 x 768): the mean
    pairwise cosine between the positions' hidden states after the last layer stays far from 1 with _fusion_sd's init."""
    from oracle import bert_fusion
    W, layers, I, Dp, vocab, max_pos = 768, 12, 3072, 256, 30524, 512
    sd = _fusion_sd(layers, W, I, 768, Dp, vocab, max_pos, seed=0)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 30522, (2, 32), generator=g, dtype=torch.int32)
    ids[:, 0] = 30523
    with torch.no_grad():
        h = bert_fusion.fusion_forward(sd, ids, torch.ones(2, 32, dtype=torch.int32), torch.randn(2, 577, 768, generator=g))
    hn = torch.nn.functional.normalize(h[0], dim=-1)
    mean_cos = ((hn @ hn.t()).sum() - 32) / (32 * 31)
    assert mean_cos < 0.7, mean_cos


@pytest.mark.parametrize("enc_width", [768, 1024])
def test_blip_fusion_full_shape(enc_width):
    """BASELINE config 4 at the shape it names: med_config.json's BERT-base (12 layers, 768 wide, 12 heads, FFN 3072) with
    cross-attention over 577 image tokens of width `enc_width` (768 = ViT-B, 1024 = create_vit('large'),
    blip4cir/blip.py:206-212), B = 128, 32-token captions, 30 000 x 256 bank, tau 0.03.
      (1) the first captions' queries against oracle/bert_fusion.py at full model size (1 - cos <= 1e-3);
      (2) batch-permutation equivariance of the queries;
      (3) the loss recomputed by the oracle from the GPU queries;
      (4) backward linear in the incoming gradient (x2 is exact in bf16 / fp32);
      (5) at B = 8 of the same model: loss and every parameter gradient against the oracle's autograd."""
    _need_gpu()
    from oracle import bank_loss, bert_fusion
    from spn4cir_amd import ops
    from spn4cir_amd.fusion import FusionEncoder
    W, layers, heads, I, Dp, vocab, max_pos = 768, 12, 12, 3072, 256, 30524, 512
    B, L, S, M, tau = 128, 32, 577, 30000, 0.03
    sd = _fusion_sd(layers, W, I, enc_width, Dp, vocab, max_pos, seed=0)
    enc_model = FusionEncoder(W, layers, heads, I, enc_width, Dp, vocab, max_pos, "cuda")
    enc_model.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 30522, (B, L), generator=g, dtype=torch.int32)
    ids[:, 0] = 30523                                                   # [ENC] (blip_cir.py:87-88)
    lens = torch.randint(6, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int32)
    ids = ids * mask
    enc = torch.randn(B, S, enc_width, generator=g)
    bank = torch.nn.functional.normalize(torch.randn(M, Dp, generator=g))
    labels = torch.randint(0, M, (B,), generator=g)
    enc_d = enc.cuda()
    proj = enc_model.forward(ids.cuda(), mask.cuda(), enc_d).clone()
    q, qb, inv = ops.combine_l2norm_fwd(None, None, proj)
    # (1)
    q_ref = bert_fusion.fusion_query(sd, ids[:3], mask[:3], enc[:3])
    cos = torch.nn.functional.cosine_similarity(q[:3].cpu().double(), q_ref.double(), dim=-1)
    assert (1 - cos).max() < 1e-3, cos
    # (2)
    perm = torch.randperm(B, generator=g)
    proj_p = enc_model.forward(ids[perm].contiguous().cuda(), mask[perm].contiguous().cuda(), enc_d[perm.cuda()].contiguous())
    assert _rel(proj_p, proj[perm.cuda()]) < 1e-6
    # (3)
    enc_model.forward(ids.cuda(), mask.cuda(), enc_d)
    bank_b = ops.prepare_bank(bank.cuda())
    stats = ops.bank_stats_fwd(qb, bank_b, labels.cuda(), 1.0 / tau)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    ref_loss = torch.nn.functional.cross_entropy(q.cpu() @ bank.t() / tau, labels)
    assert abs(mean.item() - ref_loss.item()) < 1e-2 * max(1.0, abs(ref_loss.item()))
    # (4)
    dq = ops.bank_grad_q(qb, bank_b, labels.cuda(), 1.0 / tau, lse, 1.0 / B)[:, :Dp].contiguous()
    dproj = ops.combine_l2norm_bwd(q, inv, dq)
    g1 = enc_model.backward(dproj).clone()
    enc_model.forward(ids.cuda(), mask.cuda(), enc_d)
    g2 = enc_model.backward(2.0 * dproj)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert _rel(g2, 2.0 * g1) < 1e-6
    del g1, g2
    # (5)
    b = 8
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    qo = bert_fusion.fusion_query(params, ids[:b], mask[:b], enc[:b])
    lo = torch.nn.functional.cross_entropy(qo @ bank.t() / tau, labels[:b])
    lo.backward()
    proj = enc_model.forward(ids[:b].contiguous().cuda(), mask[:b].contiguous().cuda(), enc_d[:b].contiguous())
    q, qb, inv = ops.combine_l2norm_fwd(None, None, proj)
    stats = ops.bank_stats_fwd(qb, bank_b, labels[:b].cuda(), 1.0 / tau)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    assert abs(mean.item() - lo.item()) < 1e-2 * max(1.0, abs(lo.item()))
    dq = ops.bank_grad_q(qb, bank_b, labels[:b].cuda(), 1.0 / tau, lse, 1.0 / b)[:, :Dp].contiguous()
    grads = enc_model.backward(ops.combine_l2norm_bwd(q, inv, dq))
    views = enc_model.named_views(grads)
    bad, worst = {}, (0.0, None)
    for k, p in params.items():
        ref = p.grad
        if k == "embeddings.position_embeddings.weight":
            ref, got = ref[:L], views[k][:L].cpu()          # rows beyond the caption length get no gradient
        else:
            got = views[k].cpu()
        if k.endswith(".self.key.bias"):
            # softmax is invariant to a shift common to all keys of a query: d loss / d key.bias is exactly zero, both
            # sides hold rounding noise - compare it with the scale of the value bias gradient instead of with itself
            vref = params[k.replace(".key.bias", ".value.bias")].grad.norm()
            assert got.norm() < 1e-2 * vref and ref.norm() < 1e-2 * vref, k
            continue
        e = _rel(got, ref)
        worst = max(worst, (e, k))
        if not e < 3e-2:            # observed worst 1.4e-2 / 1.5e-2 (enc_width 768 / 1024, query weights of layers 10 / 11): gate 2 x
            bad[k] = e
    print(f"blip fusion enc_width {enc_width}: worst gradient error {worst[0]:.3e} ({worst[1]})")
    assert not bad, bad


def test_fp8_bank_trainer_step_100k():
    """BASELINE config 5's bank shape on the trainer: Stage2Trainer.set_banks(bank_dtype="fp8") with a 100 000 x 768 bank
    (e4m3 + one scale per row) and ViT-L/14's text tower: one fused step (zscir/train_bank.py's loop body) - the loss and
    every gradient against the oracle run on the DEQUANTISED bank (the quantiser itself is pinned bit for bit in
    test_kernels_gpu.py::test_bank_fp8), and the step moves the parameters as torch's AdamW does."""
    _need_gpu()
    from oracle import optim as ooptim
    from spn4cir_amd import ops, synthetic
    from spn4cir_amd.models import CIRPlus
    from spn4cir_amd.trainer import Stage2Trainer
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    B, M, tau, lr = 8, 100000, 0.02, 2e-5
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    ids = synthetic.token_ids(B, seed=1)
    target, refer = synthetic.banks(M, D, seed=2)
    ridx, labels = synthetic.triplet_indices(B, M, seed=4)
    model = CIRPlus({k: v.clone() for k, v in sd.items()}, tau=tau, device=torch.device("cuda"), plus=True)
    tr = Stage2Trainer(model, lr=lr)
    tr.set_banks(refer, target, bank_dtype="fp8")
    assert isinstance(tr._bank, ops.Fp8Bank) and tr._bank.data.dtype == torch.uint8 and tr._bank.shape[0] == M
    deq = tr._bank.dequantize()[:, :D].cpu()
    assert (deq - target).abs().max() < 2.0 ** -4 * target.abs().max()          # e4m3: 3 mantissa bits
    loss = tr.step(ids.cuda(), ridx.cuda(), labels.cuda())
    grads = model.tower.grads.clone()
    torch.cuda.synchronize()
    f_ref, loss_ref, g_ref = _oracle_text_step(sd, ids, refer, ridx, deq, labels, tau)
    assert abs(loss.item() - loss_ref) < 1e-2 * max(1.0, abs(loss_ref))
    views = model.tower.named_views(grads)
    errs = {k: _rel(views[k].cpu(), r) for k, r in g_ref.items()}
    wk = max(errs, key=errs.get)
    print(f"fp8 bank trainer step: worst gradient error {errs[wk]:.3e} ({wk})")
    bad = {k: v for k, v in errs.items() if not v < GRAD_GATE_VITL14}
    assert not bad, bad
    # the update: torch.optim.AdamW semantics (train_negplus.py:77-83 hyper-parameters) applied to the step's own gradient
    # (a first Adam step is ~ lr * sign(g): comparing against the oracle's gradient would only re-test sign noise)
    for k in ("text_projection", "transformer.resblocks.5.mlp.c_fc.weight", "positional_embedding"):
        p = sd[k].clone()
        ooptim.adamw_step(p, views[k].cpu(), torch.zeros_like(p), torch.zeros_like(p), 1, lr)
        moved = (p - sd[k]).norm()
        assert moved > 0 and (model.tower.named_views()[k].cpu() - p).norm() < 1e-3 * moved, k


def test_config1_vitb32_inbatch_step_every_gradient():
    """BASELINE config 1 at its real size (clip4cir/train.py --wo_bank -> models.py:151-167): CLIP ViT-B/32 - text 512 x 12 x 8
    heads, vision 768 x 12 x 12 heads, patch 32, 224 x 224 images (50 tokens), D = 512 - B = 4, in-batch negatives, both towers
    trainable.  Loss and EVERY parameter gradient (spn_vision_bwd + spn_text_bwd through CIRPlus(wo_bank=True).forward /
    autograd) against the oracle's autograd on the same seeded weights and images.  The small-fixture test
    (test_model_gpu.py::test_config1_inbatch_step_matches_reference) pins the same path to the reference's own capture."""
    _need_gpu()
    from oracle import bank_loss, clip_text, clip_vision
    from spn4cir_amd import synthetic
    from spn4cir_amd.models import CIRPlus
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-B/32"]
    B, tau = 4, 0.01
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    sd.update(clip_vision.synthetic_vision_state_dict(768, 12, 32, 224, D, seed=5))
    ids = synthetic.token_ids(B, seed=1)
    g = torch.Generator().manual_seed(0)
    ref_img, tgt_img = torch.randn(B, 3, 224, 224, generator=g), torch.randn(B, 3, 224, 224, generator=g)
    model = CIRPlus(sd, tau=tau, device=torch.device("cuda"), wo_bank=True)
    out = model.forward(ids, None, None, None, refer_image=ref_img.cuda(), target_image=tgt_img.cuda())
    loss = out["bbc_loss"]
    loss.backward()
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    ref = bank_loss.inbatch_step(clip_vision.encode_image(params, ref_img), clip_text.encode_text(params, ids.long()),
                                 clip_vision.encode_image(params, tgt_img), tau)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-2 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    named = dict(model.clip.named_parameters())
    worst = {"matrix": (0.0, None), "vector": (0.0, None)}
    n = 0
    for k, p in params.items():
        if p.grad is None or float(p.grad.abs().max()) == 0.0:
            continue                                        # logit_scale-like entries the step does not touch
        gk = named[k].grad
        assert gk is not None, k
        if k == "token_embedding.weight":                   # only the rows of the batch's ids carry gradient
            rows = torch.unique(ids.long())
            err = _rel(gk.cpu()[rows], p.grad[rows])
        else:
            err = _rel(gk.cpu(), p.grad)
        kind = "matrix" if p.dim() >= 2 else "vector"
        if err > worst[kind][0]:
            worst[kind] = (err, k)
        # Weight matrices: the usual 5e-2.  Bias / LayerNorm vectors at B = 4: such a gradient is a sum over the rows that reach the
        # loss - 4 [EOS] rows, 2 x 4 class tokens - of terms that largely cancel between the reference and the target side of the
        # in-batch loss (d/dq and d/dt pull in opposite directions), so the bf16 operand rounding is measured against a small
        # remainder: observed worst 0.130 (a c_proj bias of the visual tower; 0.113 on visual.ln_post.bias: 8 rows), gate 2 x that; the
        # matrices do not cancel that way (observed worst 3.6e-2).
        assert err < (5e-2 if kind == "matrix" else 0.26), (k, err)
        n += 1
    assert n >= 12 * 12 * 2 + 8
    print(f"config 1 (ViT-B/32, B=4): loss {loss.item():.5f} oracle {ref.item():.5f}; worst gradient error over {n} tensors: "
          f"matrices {worst['matrix'][0]:.3e} ({worst['matrix'][1]}), vectors {worst['vector'][0]:.3e} ({worst['vector'][1]})")
