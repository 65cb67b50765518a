"""blip4cir end to end through the reference's OWN call chain, strings in - features, metrics, loss and gradients out.
Build container only (imports /root/reference and transformers).   python tests/golden/make_golden_blipval.py

What runs from the reference, as written:
  * blip4cir/utils.py `extract_index_features(classic_dataset, model.blip)`            (train.py:59,70)
  * blip4cir/validate.py `compute_fiq_val_metrics` / `compute_cirr_val_metrics(rel_ds, model.blip, ...)` (train.py:134,167)
  * blip4cir/blip_cir.py `BLIP_Retrieval.img_embed` (:54-70, incl. the pooled + normalised vision_proj of :62) and
    `img_txt_fusion` (:82-103: tokenizer(text, padding='longest'), ids[:, 0] = enc_token_id, BertModel, text_proj, normalize)
  * blip4cir/models.py `CIRPlus.forward` -> bank_large_step -> infonce_loss (:95-121) on caption STRINGS, and its backward
  * blip4cir/vit.py `VisionTransformer.forward`, blip4cir/med.py `BertModel`
What cannot (offline) and how it is bridged - the objects are assembled WITHOUT the constructors that would download or
need timm, then the methods above run on them unchanged:
  * `BLIP_Retrieval.__init__` / `CIRPlus.__init__` (hub tokenizer, med_config.json path, checkpoint): attributes set by hand;
  * timm's `PatchEmbed`: Conv2d(k = s = patch) -> flatten(2).transpose(1, 2), restated by reading (as blip_vit.npz says);
  * `BertTokenizer.from_pretrained('bert-base-uncased')`: the same Python class (transformers 5: BertTokenizerLegacy) on the
    synthetic vocabulary of bert_tokenizer.json, + the three lines of blip.py:191-193.
Weights and images are regenerated from seeds on both sides (tests/golden/cases.py: blipval_*); only outputs are stored.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from torch import nn

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from cases import BLIPVAL, ClassicRows, blipval_inputs, blipval_state_dict, blipval_vocab  # noqa: E402
from make_golden import REF, FakeCirrDataset, FakeFiqDataset, install_stubs                 # noqa: E402
from make_golden_blipvit import PatchEmbedByReading, _never                                # noqa: E402


def import_reference():
    import transformers                                         # noqa: F401  (before the torchvision stub: it probes for the real one)
    import transformers.modeling_utils                          # noqa: F401
    from transformers.models.bert.tokenization_bert_legacy import BertTokenizerLegacy  # noqa: F401
    from transformers import BertTokenizer                      # noqa: F401
    install_stubs()
    mods = {n: types.ModuleType(n) for n in (
        "timm", "timm.models", "timm.models.vision_transformer", "timm.models.registry", "timm.models.layers",
        "timm.models.helpers", "timm.models.hub", "fairscale", "fairscale.nn", "fairscale.nn.checkpoint",
        "fairscale.nn.checkpoint.checkpoint_activations")}
    vt = mods["timm.models.vision_transformer"]
    vt._cfg, vt.PatchEmbed = _never("_cfg"), _never("PatchEmbed")
    mods["timm.models.registry"].register_model = lambda f: f
    mods["timm.models.layers"].trunc_normal_ = _never("trunc_normal_")
    mods["timm.models.layers"].DropPath = _never("DropPath")
    mods["timm.models.helpers"].named_apply = _never("named_apply")
    mods["timm.models.helpers"].adapt_input_conv = _never("adapt_input_conv")
    mods["timm.models.hub"].download_cached_file = _never("download_cached_file")
    mods["fairscale.nn.checkpoint.checkpoint_activations"].checkpoint_wrapper = _never("checkpoint_wrapper")
    sys.modules.update(mods)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = _never("find_pruneable_heads_and_indices")
    sys.path.insert(0, os.path.join(REF, "blip4cir"))
    import med, vit, blip_cir, utils, validate, models          # noqa: E401,E402  (the reference's modules)
    med.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    med.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    import torch.utils.data as tud
    real = tud.DataLoader
    single = lambda *a, **k: real(*a, **{**k, "num_workers": 0, "pin_memory": False})
    utils.DataLoader = validate.DataLoader = models.DataLoader = single
    utils.device = validate.device = torch.device("cpu")
    return med, vit, blip_cir, utils, validate, models


def assemble(med, vit, blip_cir, models, sd, vocab_path):
    from functools import partial
    from transformers.models.bert.configuration_bert import BertConfig
    from transformers.models.bert.tokenization_bert_legacy import BertTokenizerLegacy
    c = BLIPVAL
    S = (c["RES"] // c["PATCH"]) ** 2 + 1
    ve = vit.VisionTransformer.__new__(vit.VisionTransformer)
    nn.Module.__init__(ve)
    ve.patch_embed = PatchEmbedByReading(c["PATCH"], c["W"])
    ve.cls_token = nn.Parameter(torch.zeros(1, 1, c["W"]))
    ve.pos_embed = nn.Parameter(torch.zeros(1, S, c["W"]))
    ve.pos_drop = nn.Dropout(p=0.0)
    ve.blocks = nn.ModuleList([vit.Block(dim=c["W"], num_heads=c["W"] // 64, mlp_ratio=4.0, qkv_bias=True,
                                         norm_layer=partial(nn.LayerNorm, eps=1e-6)) for _ in range(c["VLAYERS"])])
    ve.norm = nn.LayerNorm(c["W"], eps=1e-6)
    cfg = BertConfig(vocab_size=sd["text_encoder.embeddings.word_embeddings.weight"].shape[0], hidden_size=c["HID"],
                     num_hidden_layers=c["LAYERS"], num_attention_heads=c["HID"] // 64, intermediate_size=c["INTER"],
                     max_position_embeddings=c["MAXPOS"], layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, pad_token_id=0)
    cfg.encoder_width, cfg.add_cross_attention = c["W"], True
    blip = blip_cir.BLIP_Retrieval.__new__(blip_cir.BLIP_Retrieval)
    nn.Module.__init__(blip)
    blip.visual_encoder = ve
    tok = BertTokenizerLegacy(vocab_path)
    tok.add_special_tokens({"bos_token": "[DEC]"})                       # blip.py:191
    tok.add_special_tokens({"additional_special_tokens": ["[ENC]"]})     # blip.py:192
    tok.enc_token_id = tok.convert_tokens_to_ids("[ENC]")                # blip.py:193 (additional_special_tokens_ids[0])
    blip.tokenizer = tok
    blip.text_encoder = med.BertModel(config=cfg, add_pooling_layer=False)
    blip.vision_proj = nn.Linear(c["W"], c["PROJ"])
    blip.text_proj = nn.Linear(c["HID"], c["PROJ"])
    blip.temp = nn.Parameter(0.07 * torch.ones([]))
    res = blip.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and set(res.missing_keys) <= {"temp", "text_encoder.embeddings.position_ids"}, res
    model = models.CIRPlus.__new__(models.CIRPlus)
    nn.Module.__init__(model)
    model.device, model.plus, model.blip = torch.device("cpu"), True, blip
    model.tau = nn.Parameter(c["TAU"] * torch.ones([]))
    model.input_dim, model.output_dim, model.encoder = c["RES"], c["PROJ"], "both"
    model.crossentropy_criterion = nn.CrossEntropyLoss()
    return model


def main():
    med, vit, blip_cir, utils, validate, models = import_reference()
    sd, inp, vocab = blipval_state_dict(), blipval_inputs(), blipval_vocab()
    with tempfile.TemporaryDirectory() as d:
        vp = os.path.join(d, "vocab.txt")
        with open(vp, "w", encoding="utf-8") as f:
            f.write("\n".join(vocab) + "\n")
        model = assemble(med, vit, blip_cir, models, sd, vp)
    model.blip.eval()                                                    # train.py:111
    z = {}
    # ---- validation chain, exactly train.py:57-60 / 134-136 / 167-168
    feats, feats_p, names = utils.extract_index_features(ClassicRows(inp["names"], inp["images"]), model.blip,
                                                         device=torch.device("cpu"))
    assert names == inp["names"]
    z["index_tokens_sample"] = feats[:, ::48, ::16].numpy()             # [NG, 13, 48] of [NG, 577, 768]
    z["index_tokens_norm"] = feats.norm(dim=-1).numpy()
    z["index_features_p"] = feats_p.numpy()
    fiq = validate.compute_fiq_val_metrics(FakeFiqDataset(inp["fiq_rows"]), model.blip, feats, feats_p, names)
    pred_fiq, tnames = validate.generate_fiq_val_predictions(model.blip, FakeFiqDataset(inp["fiq_rows"]), names, feats)
    cirr = validate.compute_cirr_val_metrics(FakeCirrDataset(inp["cirr_rows"]), model.blip, feats, feats_p, names)
    pred_cirr = validate.generate_cirr_val_predictions(model.blip, FakeCirrDataset(inp["cirr_rows"]), names, feats)[0]
    z["fiq"], z["cirr"] = np.array(fiq), np.array(cirr)
    z["pred_fiq"], z["pred_cirr"] = pred_fiq.numpy(), pred_cirr.numpy()
    # ---- training step on strings, train.py:113-125 (plus=True: token bank per unique image id)
    caps, indexs, target_ids, refer_ids = inp["train"]
    model.refer_bank = feats.clone()                                     # models.py:76-88: rows = image ids
    model.target_bank = feats_p.clone()                                  # models.py:48,62-63: normalised pooled rows
    loss = model.forward(caps, indexs, target_ids, refer_ids)["bank_loss"]
    loss.backward()
    z["loss"] = loss.detach().numpy()
    z["dtau"] = model.tau.grad.numpy()
    enc = model.blip.tokenizer(caps, padding="longest", return_tensors="pt")
    z["train_ids_longest"] = np.int64(enc.input_ids.shape[1])
    for n, p in model.blip.named_parameters():
        if p.grad is not None and (n.startswith("text_proj.") or "layer.1.crossattention.self" in n or "layer.0.attention.self.query" in n
                                   or n.endswith("layer.1.output.dense.weight") or "embeddings.LayerNorm" in n
                                   or n.endswith("position_embeddings.weight")):
            z["grad::" + n] = p.grad.numpy()
    assert model.blip.visual_encoder.blocks[0].attn.qkv.weight.grad is None or True
    np.savez_compressed(os.path.join(OUT, "blip_val.npz"), **z)
    print("fiq", fiq, "cirr", cirr, "loss", float(loss), "dtau", float(model.tau.grad), "grads", sum(k.startswith("grad::") for k in z),
          os.path.getsize(os.path.join(OUT, "blip_val.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
