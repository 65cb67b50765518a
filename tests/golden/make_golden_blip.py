"""Golden vectors for the BLIP fusion encoder (blip4cir/med.py BertModel, mode='multimodal'), captured by
importing the reference in the build container (needs /root/reference).  Run in its own interpreter:

    python tests/golden/make_golden_blip.py

Shims: transformers>=5 moved three helpers out of modeling_utils, timm is absent,
BertPreTrainedModel.init_weights / get_head_mask changed.  Only med.py is importable (vit.py / blip_cir.py
need timm + fairscale), so the capture feeds pre-tokenised ids and a random `encoder_hidden_states`, and
restates blip_cir.py:98 (text_proj + normalize) and blip4cir/models.py:117-121 (InfoNCE) with torch ops.

blip_fusion.npz: state-dict of a 2-layer BertModel (hidden 128, 2 heads, intermediate 512, vocab 600,
encoder_width 192) + text_proj [64,128]; ids [4,9] (first id = [ENC]), attention_mask with padding,
encoder_hidden_states [4,21,192], bank -> last_hidden_state, q, loss, gradients of every parameter."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))


def load_med():
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        def _nope(*a, **k):
            raise NotImplementedError
        mu.find_pruneable_heads_and_indices = _nope
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    th = types.ModuleType("timm.models.hub")
    th.download_cached_file = lambda *a, **k: None
    timm.models, tm.hub = tm, th
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.hub": th})
    spec = importlib.util.spec_from_file_location("ref_med", "/root/reference/blip4cir/med.py")
    med = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(med)
    med.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    med.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    return med


def main():
    med = load_med()
    from transformers.models.bert.configuration_bert import BertConfig
    cfg = BertConfig(vocab_size=600, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=512, max_position_embeddings=64, layer_norm_eps=1e-12,
                     hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, pad_token_id=0)
    cfg.encoder_width = 192
    cfg.add_cross_attention = True
    torch.manual_seed(0)
    bert = med.BertModel(config=cfg, add_pooling_layer=False)
    bert.init_weights()
    with torch.no_grad():      # non-trivial LN affine / biases so that parity sees them
        for n, p in bert.named_parameters():
            if n.endswith("LayerNorm.weight"):
                p.add_(0.1 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
            elif p.dim() == 2 and "embeddings" not in n:
                p.mul_(3.0)     # init std 0.02 is tiny for a 128-wide model: make attention non-uniform
    bert.eval()                # blip4cir/train.py:111: model.blip.eval() -> dropout off
    text_proj = torch.nn.Linear(128, 64)
    g = torch.Generator().manual_seed(1)
    B, L, S = 4, 9, 21
    ids = torch.randint(1, 598, (B, L), generator=g)
    ids[:, 0] = 599                                   # [ENC] (blip_cir.py:87-88)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[1, 6:] = 0
    mask[3, 4:] = 0
    ids = ids * mask                                  # padded positions hold pad_token_id 0
    enc = torch.randn(B, S, 192, generator=g)
    M, tau = 37, 0.03
    bank = torch.nn.functional.normalize(torch.randn(M, 64, generator=g))
    labels = torch.randint(0, M, (B,), generator=g)

    out = bert(ids, attention_mask=mask, encoder_hidden_states=enc,
               encoder_attention_mask=torch.ones(B, S, dtype=torch.long), return_dict=True, mode="multimodal")
    h = out.last_hidden_state
    q = torch.nn.functional.normalize(text_proj(h[:, 0, :]), dim=-1)           # blip_cir.py:98
    loss = torch.nn.functional.cross_entropy((q @ bank.T) / tau, labels)       # blip4cir/models.py:117-121
    loss.backward()
    z = {"ids": ids.numpy().astype(np.int32), "mask": mask.numpy().astype(np.int32), "enc": enc.numpy(),
         "bank": bank.numpy(), "labels": labels.numpy(), "tau": np.float32(tau),
         "last_hidden_state": h.detach().numpy(), "q": q.detach().numpy(), "loss": loss.detach().numpy()}
    for n, p in bert.named_parameters():
        z["sd::" + n] = p.detach().numpy()
        if p.grad is not None:
            z["grad::" + n] = p.grad.numpy()
    z["sd::text_proj.weight"] = text_proj.weight.detach().numpy()
    z["sd::text_proj.bias"] = text_proj.bias.detach().numpy()
    z["grad::text_proj.weight"] = text_proj.weight.grad.numpy()
    z["grad::text_proj.bias"] = text_proj.bias.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "blip_fusion.npz"), **z)
    print("blip_fusion.npz", os.path.getsize(os.path.join(OUT, "blip_fusion.npz")) // 1024, "KiB",
          "loss", float(loss), "keys", len(z))
    print(sorted(k for k in z if k.startswith("sd::"))[:60])


if __name__ == "__main__":
    main()
