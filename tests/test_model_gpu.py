"""GPU tests of the host protocol: CIRPlus (reference-style autograd loop), the fused trainer
against the CPU oracle, and the Recall@K surface against metrics captured from the reference."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _tiny_sd(golden_dir):
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    return z, {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}


def test_cirplus_reference_style_loop(golden_dir):
    """forward -> {'bank_loss'} with grad; loss.backward(); torch.optim.AdamW(model.parameters()).step()
    exactly as clip4cir/train_negplus.py:77-123 drives the model."""
    _need_gpu()
    from spn4cir_amd.models import CIRPlus
    z, sd = _tiny_sd(golden_dir)
    s = np.load(os.path.join(golden_dir, "cirplus_step.npz"))
    model = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"), plus=True)
    keys = set(model.state_dict().keys())
    assert "clip.transformer.resblocks.1.attn.in_proj_weight" in keys and "clip.text_projection" in keys
    assert "clip.visual.conv1.weight" in keys and "clip.logit_scale" in keys
    assert not model.clip.visual.conv1.weight.requires_grad            # frozen image tower
    model.refer_bank = torch.from_numpy(s["refer_bank"])
    model.target_bank = torch.from_numpy(s["target_bank"])
    ids = torch.from_numpy(z["ids"])
    B = ids.shape[0]
    opt = torch.optim.AdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 1e-3,
                              "betas": (0.9, 0.999), "eps": 1e-7}])
    opt.zero_grad()
    out = model.forward(ids, torch.arange(B), torch.from_numpy(s["tgt_img_ids"]), torch.from_numpy(s["ref_img_ids"]))
    loss = out["bank_loss"]
    assert loss.dim() == 0 and loss.requires_grad
    "{:05.3f}".format(loss)                                            # train_negplus.py:116-117
    assert abs(loss.item() - float(s["loss_plus"])) < 1e-2 * max(1.0, abs(float(s["loss_plus"])))
    (loss * 128.0).backward()                                          # GradScaler-style scaled loss
    g = model.clip.text_projection.grad
    ref = torch.from_numpy(s["grad_plus::text_projection"]) * 128.0
    assert ((g.cpu() - ref).norm() / ref.norm()).item() < 5e-2
    assert model.clip.logit_scale.grad is None and model.clip.visual.proj.grad is None
    before = model.clip.text_projection.detach().clone()
    for p in model.parameters():                                       # scaler.unscale_
        if p.grad is not None:
            p.grad = p.grad / 128.0
    opt.step()
    model.parameters_changed()
    assert (model.clip.text_projection.detach() - before).abs().max() > 1e-5
    loss2 = model.forward(ids, torch.arange(B), torch.from_numpy(s["tgt_img_ids"]),
                          torch.from_numpy(s["ref_img_ids"]))["bank_loss"]
    assert loss2.item() < loss.item()                                  # one step on the same batch lowers the loss
    # the packed text tower is the DEFAULT of the drop-in (ids on the host, as the DataLoader hands them over): same features bit
    # for bit and the same loss as the dense 77-position run
    assert model.pack_eot and model._pack[0] is not None and model._pack[1] < ids.numel()
    dense = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"), plus=True, pack_eot=False)
    dense.refer_bank, dense.target_bank = model.refer_bank, torch.from_numpy(s["target_bank"])
    fresh = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"), plus=True)
    fresh.refer_bank, fresh.target_bank = model.refer_bank, torch.from_numpy(s["target_bank"])
    args = (ids, torch.arange(B), torch.from_numpy(s["tgt_img_ids"]), torch.from_numpy(s["ref_img_ids"]))
    lp, ld = fresh.forward(*args)["bank_loss"], dense.forward(*args)["bank_loss"]
    assert dense._pack[0] is None and abs(lp.item() - ld.item()) < 1e-6 * max(1.0, abs(ld.item()))
    with torch.enable_grad():
        fp, fd = fresh.tower.forward(fresh.tokenize(ids), *fresh._pack).clone(), dense.tower.forward(dense.tokenize(ids)).clone()
    assert torch.equal(fp, fd)
    # per-triplet reference rows (plus=False, models_negplus.py:135)
    model2 = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"), plus=False)
    model2.refer_bank = torch.from_numpy(s["trip_bank"])
    model2.target_bank = torch.from_numpy(s["target_bank"])
    l3 = model2.forward(ids, torch.arange(B), torch.from_numpy(s["tgt_img_ids"]), torch.from_numpy(s["ref_img_ids"]))
    assert abs(l3["bank_loss"].item() - float(s["loss_trip"])) < 1e-2 * max(1.0, abs(float(s["loss_trip"])))


def test_checkpoint_roundtrip(golden_dir, tmp_path):
    _need_gpu()
    from spn4cir_amd.models import CIRPlus
    z, sd = _tiny_sd(golden_dir)
    model = CIRPlus(sd, device=torch.device("cuda"))
    ids = torch.from_numpy(z["ids"]).cuda()
    f0 = model.encode_text(ids).clone()
    path = str(tmp_path / "best.pt")
    torch.save({"epoch": 0, "state_dict": model.state_dict()}, path)   # utils.save_model format (utils.py:53-67)
    with torch.no_grad():
        model.clip.text_projection.mul_(0.5)
    model.parameters_changed()
    assert (model.encode_text(ids) - f0).abs().max() > 1e-3
    model.load_ckpt(path, is_origin=False)
    assert torch.allclose(model.encode_text(ids), f0, atol=1e-6)
    torch.save({"CLIP": sd}, str(tmp_path / "stage1.pt"))             # stage-1 format (models_negplus.py:54-55)
    model.load_ckpt(str(tmp_path / "stage1.pt"), is_origin=True)
    assert torch.allclose(model.encode_text(ids), f0, atol=1e-6)
    # a stage-1 checkpoint also replaces the image tower: the derived bf16 operands / folded weights must follow
    img = torch.from_numpy(z["image"]).cuda()
    v0 = model.encode_image(img).clone()
    sd2 = {k: (v * 0.5 if k == "visual.proj" else v) for k, v in sd.items()}
    torch.save({"CLIP": sd2}, str(tmp_path / "stage1b.pt"))
    model.load_ckpt(str(tmp_path / "stage1b.pt"), is_origin=True)
    assert torch.allclose(model.encode_image(img), 0.5 * v0, atol=1e-3 * v0.abs().max().item())
    zr = np.load(os.path.join(golden_dir, "tiny_clip_resnet.npz"))
    sdr = {k: v for k, v in sd.items() if not k.startswith("visual.")}
    sdr.update({k[4:]: torch.from_numpy(zr[k]) for k in zr.files if k.startswith("sd::")})
    mr = CIRPlus(sdr, device=torch.device("cuda"))
    imr = torch.from_numpy(zr["image"]).cuda()
    r0 = mr.encode_image(imr).clone()
    sdr2 = {k: (v * 0.5 if k == "visual.attnpool.c_proj.weight" else v) for k, v in sdr.items()}
    torch.save({"CLIP": sdr2}, str(tmp_path / "stage1r.pt"))
    mr.load_ckpt(str(tmp_path / "stage1r.pt"), is_origin=True)
    bias = torch.from_numpy(zr["sd::visual.attnpool.c_proj.bias"]).cuda()
    assert torch.allclose(mr.encode_image(imr) - bias, 0.5 * (r0 - bias), atol=1e-5)


def test_trainer_steps_match_oracle():
    """Three fused steps (fwd, bank loss, bwd, AdamW, bf16 refresh) vs the CPU oracle + torch AdamW."""
    _need_gpu()
    from oracle import bank_loss, clip_text
    from spn4cir_amd import synthetic
    from spn4cir_amd.models import CIRPlus
    from spn4cir_amd.trainer import Stage2Trainer
    W, layers, D, vocab, B, M, tau, lr = 128, 2, 128, 600, 16, 900, 0.03, 1e-3
    sd = synthetic.text_state_dict(W, layers, D, vocab=vocab, seed=0)
    target, refer = synthetic.banks(M, D)
    ids = synthetic.token_ids(B, vocab=vocab, seed=1)
    ridx, labels = synthetic.triplet_indices(B, M)
    model = CIRPlus({k: v.clone() for k, v in sd.items()}, tau=tau, device=torch.device("cuda"), plus=True)
    tr = Stage2Trainer(model, lr=lr)
    tr.set_banks(refer, target)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW([{"params": list(params.values()), "lr": lr, "betas": (0.9, 0.999), "eps": 1e-7}])
    for step in range(3):
        loss = tr.step(ids.cuda(), ridx.cuda(), labels.cuda())
        opt.zero_grad()
        ref = bank_loss.bank_large_step(refer, ridx, clip_text.encode_text(params, ids), target, labels, tau)
        ref.backward()
        opt.step()
        assert abs(loss.item() - ref.item()) < 2e-2 * max(1.0, abs(ref.item())), (step, loss.item(), ref.item())
    views = model.tower.named_views()
    for k in ("text_projection", "transformer.resblocks.0.mlp.c_fc.weight", "positional_embedding", "ln_final.weight"):
        d = (views[k].cpu() - params[k].detach())
        moved = (params[k].detach() - sd[k]).norm()
        assert d.norm() < 0.15 * moved, (k, d.norm().item(), moved.item())    # same trajectory as the oracle


class _StubModel:
    """encode_text returns pre-made features in call order (the metric code only needs the protocol)."""

    def __init__(self, feats, dim):
        self.feats, self.output_dim, self.pos = feats, dim, 0

    def encode_text(self, captions):
        out = self.feats[self.pos:self.pos + len(captions)]
        self.pos += len(captions)
        return out


def test_recall_metrics_match_reference(golden_dir):
    _need_gpu()
    from spn4cir_amd import validate
    z = np.load(os.path.join(golden_dir, "recall.npz"))
    names = json.loads(str(z["names"]))
    members = json.loads(str(z["members"]))
    gallery = torch.from_numpy(z["gallery"]).cuda()
    text = torch.from_numpy(z["text_feats"]).cuda()
    ref_idx, tgt_idx = z["ref_idx"], z["tgt_idx"]
    fiq_rows = [(names[r], names[t], [f"cap a {i}.", f"cap b {i}?"]) for i, (r, t) in enumerate(zip(ref_idx, tgt_idx))]
    r10, r50 = validate.compute_fiq_val_metrics(fiq_rows, _StubModel(text, 64), gallery, names)
    assert (r10, r50) == pytest.approx(tuple(z["fiq"]), abs=1e-9)
    cirr_rows = [(names[r], names[t], f"cap {i}", members[i]) for i, (r, t) in enumerate(zip(ref_idx, tgt_idx))]
    cirr = validate.compute_cirr_val_metrics(cirr_rows, _StubModel(text, 64), gallery, names)
    assert cirr == pytest.approx(tuple(z["cirr"]), abs=1e-4)
    pred, _, _ = validate.generate_fiq_val_predictions(_StubModel(text, 64), fiq_rows, names, gallery)
    assert np.allclose(pred.cpu().numpy(), z["pred"], atol=1e-6)


def test_validation_calls_the_combiner_seam(golden_dir):
    """validate.py:92,206 call `model.combining_function` - the Combiner API the reference's scripts rely on.  The fused
    gather + sum + normalise kernel is only a fast path for the built-in element-wise sum; any other callable put on the
    attribute (here zscir/models_bank.py:49-54's need_norm=True variant) must be what validation runs."""
    _need_gpu()
    import functools
    from spn4cir_amd import validate
    from spn4cir_amd.models_bank import CIRPlus
    z = np.load(os.path.join(golden_dir, "recall.npz"))
    names = json.loads(str(z["names"]))
    gallery = torch.from_numpy(z["gallery"]).cuda()
    text = torch.from_numpy(z["text_feats"]).cuda()
    ref_idx, tgt_idx = z["ref_idx"], z["tgt_idx"]
    rows = [(names[r], names[t], [f"cap a {i}.", f"cap b {i}?"]) for i, (r, t) in enumerate(zip(ref_idx, tgt_idx))]
    calls = []

    class Swapped(_StubModel):
        def combining_function(self, ref, txt):
            calls.append(ref.shape[0])
            return CIRPlus.element_wise_sum(self, ref, txt, need_norm=True)
    pred, _, _ = validate.generate_fiq_val_predictions(Swapped(text, 64), rows, names, gallery)
    assert sum(calls) == len(rows)
    g, t = torch.from_numpy(z["gallery"]), torch.from_numpy(z["text_feats"])
    F = torch.nn.functional
    want = F.normalize(F.normalize(g[torch.from_numpy(ref_idx).long()]) + F.normalize(t), dim=-1)
    assert np.allclose(pred.cpu().numpy(), want.numpy(), atol=1e-6)
    assert not np.allclose(want.numpy(), z["pred"], atol=1e-3)                  # it really is a different Combiner
    # the built-in sums keep the fused kernel and the reference's numbers; a bound method with the default need_norm too
    stub = _StubModel(text, 64)
    stub.combining_function = functools.partial(CIRPlus.element_wise_sum, stub)  # not marked: goes through the seam, same result
    pred2, _, _ = validate.generate_fiq_val_predictions(stub, rows, names, gallery)
    assert np.allclose(pred2.cpu().numpy(), z["pred"], atol=1e-6)
    assert getattr(CIRPlus.element_wise_sum, "_spn_fused_sum", False)


def test_vision_tower_matches_reference(golden_dir):
    """CLIP.encode_image through the HIP kernels vs the reference's fp32 CPU output (tiny ViT, patch 16)."""
    _need_gpu()
    from spn4cir_amd.models import CIRPlus
    z, sd = _tiny_sd(golden_dir)
    model = CIRPlus(sd, device=torch.device("cuda"))
    assert model.input_dim == 32 and model.vision is not None
    feats = model.encode_image(torch.from_numpy(z["image"]).cuda()).cpu()
    ref = torch.from_numpy(z["image_feats"])
    cos = torch.nn.functional.cosine_similarity(feats.double(), ref.double(), dim=-1)
    assert (1 - cos).max() < 1e-3
    assert (feats - ref).norm() / ref.norm() < 2e-2
    # a larger, ViT-L/14-shaped geometry (patch 14 -> K = 588 padded to 640, 257 tokens) against the oracle
    from oracle import clip_vision
    g = torch.Generator().manual_seed(3)
    W, layers, D, p, res = 128, 1, 64, 14, 224
    big = {k: v for k, v in sd.items() if not k.startswith("visual.")}
    big["visual.conv1.weight"] = torch.randn(W, 3, p, p, generator=g) * 0.03
    big["visual.class_embedding"] = torch.randn(W, generator=g) * 0.1
    big["visual.positional_embedding"] = torch.randn((res // p) ** 2 + 1, W, generator=g) * 0.1
    for k in ("ln_pre", "ln_post"):
        big[f"visual.{k}.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        big[f"visual.{k}.bias"] = 0.1 * torch.randn(W, generator=g)
    for k, v in sd.items():
        if k.startswith("visual.transformer.resblocks.0."):
            big[k] = v
    big["visual.proj"] = torch.randn(W, D, generator=g) * 0.1
    m2 = CIRPlus(big, device=torch.device("cuda"))
    img = torch.randn(3, 3, res, res, generator=g)
    got = m2.encode_image(img.cuda()).cpu()
    want = clip_vision.encode_image(big, img)
    cos = torch.nn.functional.cosine_similarity(got.double(), want.double(), dim=-1)
    assert (1 - cos).max() < 1e-3


class _FakeTrainSet:
    """7-tuples of the reference's relative/train CIRDataset with random 'images' (data_utils_negplus.py:265)."""

    def __init__(self, n_trip, n_img, res, seed):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.randn(n_img, 3, res, res, generator=g)
        self.ref = torch.randint(0, n_img, (n_trip,), generator=g)
        self.tgt = torch.randint(0, n_img, (n_trip,), generator=g)
        self.image_id = n_img

    def __len__(self):
        return len(self.ref)

    def __getitem__(self, i):
        r, t = int(self.ref[i]), int(self.tgt[i])
        return self.images[r], f"cap {i}", self.images[t], i, i, r, t


def test_bank_builders(golden_dir, tmp_path):
    _need_gpu()
    from oracle import clip_vision
    from spn4cir_amd.models import CIRPlus
    z, sd = _tiny_sd(golden_dir)
    model = CIRPlus(sd, device=torch.device("cuda"), plus=True, neg_num=3)
    ds = _FakeTrainSet(40, 11, 32, 5)
    path = str(tmp_path / "fiq_bank.pth")
    model.extract_bank_features(ds, bank_path=path)
    refer, target = torch.load(path)                                   # the reference's file format
    assert refer.shape == (40, 64) and target.shape == (11, 64)
    want = clip_vision.encode_image(sd, ds.images)
    used = torch.unique(torch.cat([ds.ref, ds.tgt]))
    cos = torch.nn.functional.cosine_similarity(target[used].double(), want[used].double(), dim=-1)
    assert (1 - cos).min() > -1e-6 and (1 - cos).max() < 1e-3
    assert torch.allclose(target[used].norm(dim=1), torch.ones(len(used)), atol=1e-4)
    assert torch.allclose(refer, want[ds.ref], atol=0.05, rtol=0.05)
    model.extract_refer_bank_features(ds, bank_path=str(tmp_path / "fiq_refer_bank.pth"))
    assert model.refer_bank.shape == (11, 64) and model.refer_bank.is_cuda
    unl = [torch.randn(3, 32, 32) for _ in range(5)]
    model.extract_unlabeled_bank_features(unl, bank_path=str(tmp_path / "fiq_bank_unlabeled.pth"))
    assert model.M == 11 and model.target_bank.shape == (14, 64)       # neg_num = 3 of the 5 unlabeled rows kept
    # and the step runs on banks built this way
    ids = torch.from_numpy(z["ids"])
    out = model.forward(ids, torch.arange(6), ds.tgt[:6], ds.ref[:6])
    assert torch.isfinite(out["bank_loss"])


def test_config1_inbatch_step_matches_reference(golden_dir):
    """BASELINE config 1 (clip4cir/train.py --wo_bank -> models.py:151-167): in-batch negatives with the visual
    tower trainable.  Loss and the gradient of EVERY parameter (text + visual) against vectors captured from the
    reference (tests/golden/make_golden_inbatch.py).  Tolerances as for the stage-2 step: bf16 GEMM operands vs
    the reference's fp32 - loss 1e-2 relative, per-parameter gradient 5e-2 relative L2."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import numpy as np
    from spn4cir_amd.models import CIRPlus
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    s = np.load(os.path.join(golden_dir, "cirplus_inbatch.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    model = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"), wo_bank=True)
    ids = torch.from_numpy(s["ids"])
    out = model.forward(ids, None, None, None, refer_image=torch.from_numpy(s["refer_image"]).cuda(),
                        target_image=torch.from_numpy(s["target_image"]).cuda())
    loss = out["bbc_loss"]
    ref_loss = float(s["loss"])
    assert abs(loss.item() - ref_loss) < 1e-2 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    loss.backward()
    named = dict(model.clip.named_parameters())
    worst, n = 0.0, 0
    for k in s.files:
        if not k.startswith("grad::"):
            continue
        g, r = named[k[6:]].grad, torch.from_numpy(s[k])
        assert g is not None, k
        err = ((g.cpu() - r).norm() / r.norm().clamp_min(1e-12)).item()
        worst = max(worst, err)
        assert err < 5e-2, (k, err)
        n += 1
    assert n == 61
    print("config 1: loss", loss.item(), "ref", ref_loss, "worst per-parameter grad error", worst)


def test_zscir_models_bank_protocol(golden_dir):
    """zscir/models_bank.py argument order + per-triplet reference rows: reproduces the `loss_trip` capture of the
    reference (plus=False path) and its text_projection gradient."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import numpy as np
    from spn4cir_amd.models_bank import CIRPlus
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    s = np.load(os.path.join(golden_dir, "cirplus_step.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    model = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"))
    model.refer_bank = torch.from_numpy(s["trip_bank"])
    model.target_bank = torch.from_numpy(s["target_bank"])
    B = z["ids"].shape[0]
    out = model.forward(None, torch.from_numpy(z["ids"]), None, torch.arange(B), torch.from_numpy(s["tgt_img_ids"]),
                        torch.from_numpy(s["ref_img_ids"]), grad_ckpt=True)
    loss = out["bank_loss"]
    ref = float(s["loss_trip"])
    assert abs(loss.item() - ref) < 1e-2 * max(1.0, abs(ref))
    loss.backward()
    g = dict(model.clip.named_parameters())["text_projection"].grad.cpu()
    r = torch.from_numpy(s["grad_trip_text_projection"])
    assert ((g - r).norm() / r.norm()).item() < 5e-2


def test_exact_encode_mode_matches_fp32_oracle(golden_dir):
    """fp32-exact towers (f32-input MFMA, fp32 attention): features against the REFERENCE's own fp32 outputs captured in
    tiny_clip.npz - 1e-5 relative to the largest feature (accumulation order is the only difference) - and, on a
    synthetic retrieval task, top-K index sets identical to the fp32 oracle end to end."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import numpy as np
    from oracle import clip_text, recall
    from spn4cir_amd.models import CIRPlus
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    model = CIRPlus(sd, device=torch.device("cuda"), exact_eval=True)
    with torch.no_grad():
        tf = model.encode_text(torch.from_numpy(z["ids"])).cpu()
        vf = model.encode_image(torch.from_numpy(z["image"]).cuda()).cpu()
    tref, vref = torch.from_numpy(z["text_feats"]), torch.from_numpy(z["image_feats"])
    assert (tf - tref).abs().max() < 1e-5 * tref.abs().max().clamp_min(1.0), (tf - tref).abs().max()
    assert (vf - vref).abs().max() < 1e-5 * vref.abs().max().clamp_min(1.0), (vf - vref).abs().max()
    # end-to-end ranking identity on a ViT-B/32-sized text tower
    from spn4cir_amd import synthetic
    from spn4cir_amd.text_tower import TextTower
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-B/32"]
    sd2 = synthetic.text_state_dict(W, layers, D, seed=0)
    ids = synthetic.token_ids(24, seed=1)
    tower = TextTower(W, layers, heads, D, 49408, 77, "cuda")
    tower.load_clip_state_dict(sd2)
    t_gpu = tower.forward_exact(ids.cuda()).cpu()
    with torch.no_grad():
        t_cpu = clip_text.encode_text(sd2, ids)
    assert (t_gpu - t_cpu).abs().max() < 2e-5 * t_cpu.abs().max()
    g = torch.Generator().manual_seed(9)
    gallery = torch.nn.functional.normalize(torch.randn(6000, D, generator=g))
    ref = torch.randn(24, D, generator=g)
    og, _ = recall.ranked_indices(torch.nn.functional.normalize(ref + t_gpu).numpy(), gallery.numpy())
    oc, _ = recall.ranked_indices(torch.nn.functional.normalize(ref + t_cpu).numpy(), gallery.numpy())
    same10 = np.mean([set(og[i, :10]) == set(oc[i, :10]) for i in range(24)])
    same50 = np.mean([set(og[i, :50]) == set(oc[i, :50]) for i in range(24)])
    print("exact mode: top-10 / top-50 set identity", same10, same50)
    assert same10 == 1.0 and same50 >= 0.95


def test_resnet_tower_matches_reference(golden_dir):
    """CLIP ModifiedResNet image tower (fp32 path: im2col + f32-MFMA GEMM with folded BatchNorm, attention pool)
    against the reference's encode_image on a tiny RN CLIP: 1e-4 relative to the largest feature (BatchNorm folding
    and accumulation order), 1 - cos < 1e-6."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import numpy as np
    from spn4cir_amd.resnet_tower import ResNetTower
    z = np.load(os.path.join(golden_dir, "tiny_clip_resnet.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    t = ResNetTower(sd, "cuda")
    assert (t.layers, t.width, t.res, t.embed_dim) == ((1, 2, 1, 1), 8, 64, 128)
    out = t.forward(torch.from_numpy(z["image"])).cpu()
    ref = torch.from_numpy(z["image_feats"])
    assert (out - ref).abs().max() < 1e-4 * ref.abs().max(), (out - ref).abs().max()
    cos = torch.nn.functional.cosine_similarity(out.double(), ref.double(), dim=-1)
    assert (1 - cos).max() < 1e-6


def test_resnet_tower_fast_path(golden_dir):
    """ModifiedResNet bf16 throughput mode (channels padded to 64, bf16 MFMA convolutions, csrc/resnet.hip) against the
    reference's encode_image on the tiny RN CLIP and against the tower's own fp32 path on a wider synthetic one:
    1 - cos < 2e-3 (bf16 activations through every convolution), forward_exact() still gives the fp32 result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import sys
    import numpy as np
    from spn4cir_amd.resnet_tower import ResNetTower
    z = np.load(os.path.join(golden_dir, "tiny_clip_resnet.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    t = ResNetTower(sd, "cuda", fast=True)
    img = torch.from_numpy(z["image"])
    ref = torch.from_numpy(z["image_feats"])
    out = t.forward(img).cpu()
    cos = torch.nn.functional.cosine_similarity(out.double(), ref.double(), dim=-1)
    assert (1 - cos).max() < 2e-3, (1 - cos).max()
    assert (t.forward_exact(img).cpu() - ref).abs().max() < 1e-4 * ref.abs().max()
    # a wider geometry (odd widths exercise the channel padding: 24 -> 64, 96 -> 128, ...), strides and downsampling
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from vision_bench import resnet_state_dict
    sd2 = resnet_state_dict((2, 2, 2, 2), 24, 96, 256, seed=3)
    t2 = ResNetTower(sd2, "cuda", fast=True)
    img2 = torch.randn(3, 3, 96, 96, generator=torch.Generator().manual_seed(5))
    a, b = t2.forward(img2).cpu(), t2.forward_exact(img2).cpu()
    cos2 = torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)
    assert (1 - cos2).max() < 2e-3, (1 - cos2).max()


def test_cirplus_with_resnet_image_tower(golden_dir):
    """A checkpoint whose visual tower is a ModifiedResNet (train_negplus.py's default RN50x4 family): encode_image
    goes through ResNetTower, input_dim follows the attention pool's grid."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import os
    import numpy as np
    from spn4cir_amd.models import CIRPlus
    zt = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    zr = np.load(os.path.join(golden_dir, "tiny_clip_resnet.npz"))
    sd = {k[4:]: torch.from_numpy(zt[k]) for k in zt.files if k.startswith("sd::") and not k.startswith("sd::visual.")}
    sd.update({k[4:]: torch.from_numpy(zr[k]) for k in zr.files if k.startswith("sd::")})
    model = CIRPlus(sd, device=torch.device("cuda"))
    assert model.input_dim == 64
    ref = torch.from_numpy(zr["image_feats"])
    out = model.encode_image(torch.from_numpy(zr["image"]).cuda()).cpu()          # bf16 throughput mode (as the ViT tower)
    assert (1 - torch.nn.functional.cosine_similarity(out.double(), ref.double(), dim=-1)).max() < 2e-3
    model.exact_eval = True                                                       # fp32 path
    out = model.encode_image(torch.from_numpy(zr["image"]).cuda()).cpu()
    assert (out - ref).abs().max() < 1e-4 * ref.abs().max()


def test_fusion_validation_matches_reference(golden_dir):
    """spn4cir_amd.validate_fusion vs tgcir/validate.py (= blip4cir/validate.py) on a synthetic token gallery: same
    FashionIQ / CIRR metrics (reference kept in the FashionIQ ranking, dropped in CIRR; capitalised caption join)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cases import StubFusion, valfusion_inputs
    from spn4cir_amd import validate_fusion as vf
    z = np.load(os.path.join(golden_dir, "valfusion.npz"))
    tokens, pooled, names, q, fiq_rows, cirr_rows = valfusion_inputs()
    r10, r50 = vf.compute_fiq_val_metrics(fiq_rows, StubFusion(q), tokens, pooled, names)
    assert abs(r10 - float(z["fiq"][0])) < 1e-4 and abs(r50 - float(z["fiq"][1])) < 1e-4
    cirr = vf.compute_cirr_val_metrics(cirr_rows, StubFusion(q), tokens, pooled, names)
    assert np.allclose(np.array(cirr), z["cirr"], atol=1e-4)

    class Blip(StubFusion):                      # blip4cir's call shape: (r_image_embeds, t_image_embeds, text)
        def img_txt_fusion(self, r, t, text, train=False):
            assert t is None
            return StubFusion.img_txt_fusion(self, r, text)
    r10b, r50b = vf.compute_fiq_val_metrics(fiq_rows, Blip(q), tokens, pooled, names)
    assert (r10b, r50b) == (r10, r50)
    # the capitalised join reaches the model (blip4cir/validate.py:92-94)
    seen = []

    class Spy(StubFusion):
        def img_txt_fusion(self, r, mod):
            seen.extend(mod)
            return StubFusion.img_txt_fusion(self, r, mod)
    vf.generate_fiq_val_predictions(Spy(q), fiq_rows[:2], names, tokens)
    assert seen == ["Cap a 0 and cap b 0", "Cap a 1 and cap b 1"]


def test_full_size_step_properties():
    """BASELINE config 2 at full size (ViT-L/14 text tower, B = 256, 77 tokens, 40 000 x 768 bank) through properties
    that do not need a full-size CPU run: batch-permutation equivariance, gradient linearity, the loss recomputed by
    the oracle from the GPU features, and the first captions' features against the oracle tower."""
    _need_gpu()
    from oracle import bank_loss, clip_text
    from spn4cir_amd import ops, synthetic
    from spn4cir_amd.text_tower import TextTower
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    B, M, tau = 256, 40000, 0.02
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    t = TextTower(W, layers, heads, D, device="cuda")
    t.load_clip_state_dict(sd)
    ids = synthetic.token_ids(B, seed=1)
    target, refer = synthetic.banks(M, D, seed=2)
    ridx, labels = synthetic.triplet_indices(B, M, seed=4)
    feats = t.forward(ids.cuda()).clone()
    # (1) a caption's feature does not depend on its position in the batch (every row runs the same k order)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    feats_p = t.forward(ids[perm].contiguous().cuda())
    assert torch.equal(feats_p, feats[perm.cuda()])
    # (2) the first captions against the oracle tower at full model size (north_star: 1 - cos <= 1e-3)
    ref4 = clip_text.encode_text(sd, ids[:4].long())
    cos = torch.nn.functional.cosine_similarity(feats[:4].cpu().double(), ref4.double(), dim=-1)
    assert (1 - cos).max() < 1e-3
    # (3) loss: kernels vs the oracle applied to the same GPU features (bf16 operand rounding only)
    t.forward(ids.cuda())
    q, qb, inv = ops.combine_l2norm_fwd(refer.cuda(), ridx.cuda(), feats)
    bank_b = ops.prepare_bank(target.cuda())
    stats = ops.bank_stats_fwd(qb, bank_b, labels.cuda(), 1.0 / tau)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    ref_loss = bank_loss.bank_large_step(refer, ridx, feats.cpu(), target, labels, tau)
    assert abs(mean.item() - ref_loss.item()) < 1e-2 * max(1.0, abs(ref_loss.item()))
    # (4) backward is linear in the incoming gradient: doubling d(feats) doubles every parameter gradient exactly
    #     (powers of two survive the bf16 roundings of the intermediate gradients)
    dq = ops.bank_grad_q(qb, bank_b, labels.cuda(), 1.0 / tau, lse, 1.0 / B)[:, :D].contiguous()
    dfe = ops.combine_l2norm_bwd(q, inv, dq)
    g1 = t.backward(dfe).clone()
    t.forward(ids.cuda())
    g2 = t.backward(2.0 * dfe)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert ((g2 - 2.0 * g1).norm() / (2.0 * g1).norm()).item() < 1e-6


class _Box(torch.nn.Module):
    """Container used to build a TorchScript archive with CLIP's parameter names (the published CLIP files are such
    archives, clip/clip.py:127-131)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x


def test_clip_file_formats(golden_dir, tmp_path, monkeypatch):
    """`clip_model_name` as clip.load accepts it (clip/clip.py:120-137): a TorchScript archive, a saved state dict, or a
    model NAME whose file sits in the cache directory - all give the same model."""
    _need_gpu()
    from spn4cir_amd.models import CIRPlus
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    root = _Box()
    for k, v in sd.items():
        node, parts = root, k.split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, _Box())
            node = getattr(node, p)
        node.register_parameter(parts[-1], torch.nn.Parameter(v.clone(), requires_grad=False))
    for name, val in (("input_resolution", 32), ("context_length", 77), ("vocab_size", 512)):   # clip/model.py:434-436
        root.register_buffer(name, torch.tensor(val))
    cache = tmp_path / "cache"
    cache.mkdir()
    torch.jit.script(root).save(str(cache / "ViT-B-32.pt"))
    torch.save(sd, str(tmp_path / "sd.pt"))
    ids = torch.from_numpy(z["ids"])[:4]
    dev = torch.device("cuda")
    ref = CIRPlus(sd, device=dev).encode_text(ids)
    for spec in (str(cache / "ViT-B-32.pt"), str(tmp_path / "sd.pt")):
        m = CIRPlus(spec, device=dev)
        assert torch.equal(m.encode_text(ids), ref)
        assert "input_resolution" not in m.state_dict()
    monkeypatch.setenv("SPN_CLIP_CACHE", str(cache))
    assert torch.equal(CIRPlus("ViT-B/32", device=dev).encode_text(ids), ref)
    with pytest.raises(RuntimeError, match="not found"):
        CIRPlus("ViT-L/14", device=dev)                         # named, but its file is not in the cache


def _loop_model(golden_dir):
    from spn4cir_amd.models import CIRPlus
    z, sd = _tiny_sd(golden_dir)
    s = np.load(os.path.join(golden_dir, "cirplus_step.npz"))
    model = CIRPlus(sd, tau=float(s["tau"]), device=torch.device("cuda"), plus=True)
    model.refer_bank = torch.from_numpy(s["refer_bank"])
    model.target_bank = torch.from_numpy(s["target_bank"])
    ids = torch.from_numpy(z["ids"])
    args = (ids, torch.arange(ids.shape[0]), torch.from_numpy(s["tgt_img_ids"]), torch.from_numpy(s["ref_img_ids"]))
    return model, args, s


def test_grad_accumulates_like_autograd(golden_dir):
    """p.grad is a slice of the tower's overwrite-on-backward flat buffer, yet `.backward()` must ADD to an existing
    .grad as autograd does: two backward passes without zero_grad (gradient accumulation) give g1 + g2, and
    `zero_grad(set_to_none=False)` - the default of the reference's torch 1.13 - followed by a backward gives g."""
    _need_gpu()
    model, args, s = _loop_model(golden_dir)
    ref = torch.from_numpy(s["grad_plus::text_projection"])
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.0)
    model.forward(*args)["bank_loss"].backward()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert ((g1["clip.text_projection"].cpu() - ref).norm() / ref.norm()).item() < 5e-2
    (2.0 * model.forward(*args)["bank_loss"]).backward()               # accumulate: g + 2 g
    for n, p in model.named_parameters():
        if n in g1:
            assert torch.allclose(p.grad, 3.0 * g1[n], rtol=1e-4, atol=1e-6 * g1[n].abs().max().item()), n
    opt.zero_grad(set_to_none=False)                                   # keeps the (aliasing) tensors, zero-filled
    assert all(p.grad is not None and not p.grad.any() for p in params if p.grad is not None)
    model.forward(*args)["bank_loss"].backward()
    for n, p in model.named_parameters():
        if n in g1:
            assert torch.allclose(p.grad, g1[n], rtol=1e-4, atol=1e-6 * g1[n].abs().max().item()), n
    # a caller-owned .grad (not a slice of the flat buffer) accumulates too, and a mixed state is handled per tensor
    tp = model.clip.text_projection
    tp.grad = torch.ones_like(tp)
    model.clip.ln_final.weight.grad = None
    model.forward(*args)["bank_loss"].backward()
    assert torch.allclose(tp.grad, 1.0 + g1["clip.text_projection"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(model.clip.ln_final.weight.grad, g1["clip.ln_final.weight"], rtol=1e-4, atol=1e-6)
    k = "clip.transformer.resblocks.0.attn.in_proj_weight"
    assert torch.allclose(dict(model.named_parameters())[k].grad, 2.0 * g1[k], rtol=1e-4, atol=1e-6 * g1[k].abs().max().item())


def test_external_optimizer_step_is_noticed_without_parameters_changed(golden_dir):
    """The reference loop (train_negplus.py:121-123) knows nothing about parameters_changed(): an in-place optimizer
    update of the exposed nn.Parameters must reach the bf16 GEMM operands of the next forward by itself."""
    _need_gpu()
    model, args, s = _loop_model(golden_dir)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, eps=1e-7)
    l0 = model.forward(*args)["bank_loss"]
    l0.backward()
    opt.step()                                                          # no model.parameters_changed()
    opt.zero_grad()
    l1 = model.forward(*args)["bank_loss"]
    assert l1.item() < l0.item() - 1e-4, (l0.item(), l1.item())
    f1 = model.encode_text(args[0]).clone()
    with torch.no_grad():
        model.clip.text_projection.mul_(0.5)                            # any in-place write through a view
    assert torch.allclose(model.encode_text(args[0]), 0.5 * f1, atol=2e-2 * f1.abs().max().item())
    assert not model.tower.is_stale()                                   # and an unchanged model is not refreshed again


def test_fused_adamw_arithmetic_matches_torch():
    """spn4cir_amd.optim.AdamW against torch.optim.AdamW on identical gradients: parameters that are consecutive slices of
    one flat buffer with gradients that are the matching slices of another (one fused launch), the same parameters in two
    groups with broken runs (more launches), and gradients that live elsewhere (per-tensor path) - four steps each,
    with the GradScaler hand-over (optimizer.grad_scale / found_inf) from the second step on and an overflow on the SECOND
    step, followed by normal steps: a skipped step must not advance the bias correction (torch does not call
    optimizer.step() on overflow)."""
    _need_gpu()
    from spn4cir_amd import optim as spn_optim
    hp = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01)
    shapes = [(37, 16), (16,), (64, 64), (5,), (128, 3)]
    n = sum(int(np.prod(sh)) for sh in shapes)
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g)) for _ in range(4)]
    overflow_at = 1

    def build(kind):
        flat, gflat = p0.clone().cuda(), torch.zeros(n, device="cuda")
        ps, off = [], 0
        for sh in shapes:
            k = int(np.prod(sh))
            p = torch.nn.Parameter(flat[off:off + k].view(sh))
            ps.append((p, gflat[off:off + k].view(sh)))
            off += k
        plist = [p for p, _ in ps]
        if kind == "torch":
            opt = torch.optim.AdamW(plist, **hp)
        elif kind == "fused":
            opt = spn_optim.AdamW(plist, **hp)
        elif kind == "groups":
            opt = spn_optim.AdamW([{"params": plist[0::2]}, {"params": plist[1::2], "lr": hp["lr"]}], **hp)
        else:
            opt = spn_optim.AdamW(plist, **hp)
        return flat, gflat, ps, opt

    out = {}
    for kind in ("torch", "fused", "groups", "elsewhere"):
        flat, gflat, ps, opt = build(kind)
        for it, gr in enumerate(grads):
            scale = torch.tensor([1.0 if it == 0 else 512.0], device="cuda")
            found = torch.tensor([1.0 if it == overflow_at else 0.0], device="cuda")
            gflat.copy_(gr.cuda() * scale)
            for p, gv in ps:
                p.grad = gv.clone() if kind == "elsewhere" else gv
            if kind == "torch":
                if it == overflow_at:
                    continue                                 # overflow: GradScaler skips the step
                for p, _ in ps:
                    p.grad = p.grad / scale
                opt.step()
            else:
                opt.grad_scale, opt.found_inf = scale, found
                opt.step()
                del opt.grad_scale, opt.found_inf
        out[kind] = flat.clone()
        if kind != "torch":
            assert float(opt.state[ps[0][0]]["step"]) == 3.0    # applied steps only, as torch counts them
        if kind == "fused":
            assert len(opt._runs[0]) == 1                    # one flat run -> one launch per step
            assert flat._version > 0                         # the buffer's version counter moved (staleness detection)
    for kind in ("fused", "groups", "elsewhere"):
        assert (out[kind] - out["torch"]).abs().max().item() < 2e-6, kind
    assert (out["torch"] - p0.cuda()).abs().max().item() > 1e-3
    # state_dict round trip: a fresh optimizer that loads the state after step 1 continues like the uninterrupted one
    flat, gflat, ps, opt = build("fused")
    gflat.copy_(grads[0].cuda())
    for p, gv in ps:
        p.grad = gv
    opt.step()
    sd = opt.state_dict()
    flat2, gflat2, ps2, opt2 = build("fused")
    flat2.copy_(flat)
    opt2.load_state_dict(sd)
    for f_, g_, pp, o_ in ((flat, gflat, ps, opt), (flat2, gflat2, ps2, opt2)):
        g_.copy_(grads[1].cuda())
        for p, gv in pp:
            p.grad = gv
        o_.step()
    assert torch.equal(flat, flat2)


def test_fused_adamw_in_the_reference_loop(golden_dir):
    """spn4cir_amd.optim.AdamW in the reference's loop (train_negplus.py:77-84, 107-123: GradScaler, scaler.step(optimizer)):
    the loss trajectory of torch.optim.AdamW, the update reaches the next forward without parameters_changed(), the text
    tower's parameters form one flat run, and the torch-style per-parameter state is exposed."""
    _need_gpu()
    from spn4cir_amd import optim as spn_optim
    hp = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01)
    results = {}
    for name in ("torch", "fused"):
        model, args, s = _loop_model(golden_dir)
        ps = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(ps, **hp) if name == "torch" else spn_optim.AdamW(ps, **hp)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = model.forward(*args)["bank_loss"]
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(loss.item())
        results[name] = (losses, model.tower.params.clone(), opt, len(ps))
    lt, pt, _, _ = results["torch"]
    lf, pf, opt, nps = results["fused"]
    assert max(abs(a - b) for a, b in zip(lt, lf)) < 2e-3 * max(1.0, abs(lt[0])), (lt, lf)
    assert lf[2] < lf[0] - 1e-4                                # the updates reached the following forward passes
    # Elementwise the two runs agree to 1 ulp after the first step (test_fused_adamw_arithmetic_matches_torch holds the
    # exact comparison on identical gradients); from there a 1-ulp difference can flip a bf16 rounding of a weight, and
    # AdamW moves an element by ~lr whatever its gradient's size, so single elements drift by up to 2 lr per step on this
    # tiny, fast-collapsing problem: only the bulk is compared here
    diff = (pf - pt).abs()
    assert diff.max().item() < 6.5e-3 and diff.mean().item() < 2e-5
    assert sum(len(r) for r in opt._runs) <= 2
    assert len(opt.state_dict()["state"]) == nps


def test_validation_uses_exact_tower_and_keeps_pending_backward(golden_dir):
    """validate.compute_*_val_metrics as the reference's loop calls them (no torch.no_grad() around the call,
    train_negplus.py:128): CIRPlus(exact_eval=True) must take its fp32 tower there, and the training activations of
    a forward whose backward is still pending must survive the validation pass."""
    _need_gpu()
    from spn4cir_amd import validate
    from spn4cir_amd.models import CIRPlus
    z, sd = _tiny_sd(golden_dir)
    model = CIRPlus(sd, device=torch.device("cuda"), exact_eval=True)
    ids = torch.from_numpy(z["ids"])
    D = model.output_dim
    g = torch.Generator().manual_seed(5)
    gallery = torch.randn(40, D, generator=g).cuda()
    ridx = torch.arange(ids.shape[0]).cuda()
    exact = model.tower.forward_exact(ids.cuda().to(torch.int32).contiguous())
    pred = validate._predict(model, ids, ridx, gallery)                 # grad mode is ON here
    want = torch.nn.functional.normalize(gallery[ridx] + exact)
    assert (pred - want).abs().max() < 1e-6
    bf = torch.nn.functional.normalize(gallery[ridx] + model.tower.forward(ids.cuda().to(torch.int32).contiguous()))
    assert (bf - want).abs().max() > 1e-6                               # the bf16 tower is measurably different


def test_out_of_range_indices_are_loud(golden_dir):
    """refer_bank[refer_indexs] raises IndexError in the reference (models_negplus.py:133).  Host index tensors raise
    the same here; device tensors are never dereferenced out of range - the affected rows (and the loss) turn NaN."""
    _need_gpu()
    from spn4cir_amd import ops
    model, args, s = _loop_model(golden_dir)
    ids, idx, tgt, ref = args
    bad = ref.clone()
    bad[1] = model.refer_bank.shape[0]
    with pytest.raises(IndexError):
        model.forward(ids, idx, tgt, bad)
    badt = tgt.clone()
    badt[0] = -1
    with pytest.raises(IndexError):
        model.forward(ids, idx, badt, ref)
    text = torch.randn(4, 64, device="cuda")
    bank = torch.randn(10, 64, device="cuda")
    ridx = torch.tensor([0, 10, 9, -1], device="cuda")
    q, qb, inv = ops.combine_l2norm_fwd(bank, ridx, text)
    assert torch.isnan(q[1]).all() and torch.isnan(q[3]).all() and torch.isnan(inv[[1, 3]]).all()
    assert torch.isnan(qb[1, :64].float()).all() and not qb[1, 64:].float().any()
    want = torch.nn.functional.normalize(bank[[0, 9]] + text[[0, 2]])
    assert torch.allclose(q[[0, 2]], want, atol=1e-6)
