"""Golden vectors for the TG-CIR second-stage step: tgcir/models.py CIRPlus.forward ->
bank_large_step on CPU, with a tiny CLIP whose text tower and the TG-CIR head come from seeded generators that the
tests re-run (oracle.clip_text.synthetic_text_state_dict, oracle.tgcir_head.synthetic_head), so only inputs, the loss,
the query features and gradient summaries are stored.

Build container only (imports /root/reference).  The reference hard-codes .cuda() / device='cuda'
(models.py:45,56,104,138): Tensor.cuda is patched to the identity and clip.load to a CPU tiny CLIP.

    python tests/golden/make_golden_tgcir.py   ->  tests/golden/tgcir_step.npz
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, OUT)
sys.path.insert(0, ROOT)
from make_golden import REF, install_stubs  # noqa: E402

from cases import TGCIR, tgcir_image_side, tgcir_inputs, tgcir_weights  # noqa: E402

B, L, C, VOCAB, LAYERS, M, TAU, SAMPLE = (TGCIR[k] for k in ("B", "L", "C", "VOCAB", "LAYERS", "M", "TAU", "SAMPLE"))


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "tgcir"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    import clip                      # noqa: E402  (tgcir/clip)
    from clip.model import CLIP      # noqa: E402

    text_sd, head = tgcir_weights()
    torch.manual_seed(0)
    tiny = CLIP(C, 32, 2, 768, 16, L, VOCAB, C, 8, LAYERS).float()     # vision side: whatever the constructor draws
    tiny.load_state_dict(text_sd, strict=False)
    vsd, ihead, images = tgcir_image_side()
    missing = tiny.load_state_dict(vsd, strict=False)
    assert not [k for k in missing.unexpected_keys], missing.unexpected_keys
    clip.load = lambda name, device=None, jit=False: (tiny, None)

    sys.modules.pop("data_utils", None)
    sys.modules.pop("utils", None)
    import models                    # noqa: E402  (tgcir/models.py)
    model = models.CIRPlus("tiny", tau=TAU, device=torch.device("cpu"), plus=True)
    bb = model.backbone
    with torch.no_grad():
        bb.text_fc.weight.copy_(head["text_fc.weight"]); bb.text_fc.bias.copy_(head["text_fc.bias"])
        for s in range(8):
            conv = bb.tokenlearn_text.tokenizers[s].conv[0]
            conv.weight.copy_(head["tokenlearn_text.weight"][s].reshape(1, C, 1))
            conv.bias.copy_(head["tokenlearn_text.bias"][s:s + 1])
        bb.masks_text.weight.copy_(head["masks_text.weight"])
        model.s_remain_map[0].weight.copy_(head["s_remain_map.0.weight"]); model.s_remain_map[0].bias.copy_(head["s_remain_map.0.bias"])
        model.s_remain_map[2].weight.copy_(head["s_remain_map.2.weight"]); model.s_remain_map[2].bias.copy_(head["s_remain_map.2.bias"])
        bb.fc.weight.copy_(ihead["fc.weight"]); bb.fc.bias.copy_(ihead["fc.bias"])
        for s in range(8):
            conv = bb.tokenlearn.tokenizers[s].conv[0]
            conv.weight.copy_(ihead["tokenlearn.weight"][s].reshape(1, C, 1))
            conv.bias.copy_(ihead["tokenlearn.bias"][s:s + 1])
        bb.masks.weight.copy_(ihead["masks.weight"])
    ids, ref, bank, labels = tgcir_inputs()
    clip.tokenize = lambda text, *a, **k: ids.long()
    model.refer_bank = ref.clone()          # plus=True: rows picked by refer_indexs
    model.target_bank = bank.clone()
    model.train()
    model.zero_grad()
    ridx = torch.arange(B)
    loss = model.forward(["x"] * B, None, labels, ridx)["bank_loss"]
    loss.backward()
    with torch.no_grad():
        mod = bb.extract_text_fea(["x"] * B)
        q = model.img_txt_fusion(ref, ["x"] * B)
        img_tokens, img_pooled = model.img_embed(images, return_pool_and_normalized=True)
    out = {"loss": loss.detach().numpy(), "q": q.numpy(), "mod_token": mod.numpy(),
           "img_tokens": img_tokens.numpy(), "img_pooled": img_pooled.numpy()}

    def put(name, g):
        g = g.detach().reshape(-1)
        out["gnorm::" + name] = np.float64(g.double().norm().item())
        out["grad::" + name] = (g if g.numel() <= 8192 else g[::SAMPLE]).numpy().copy()

    put("text_fc.weight", bb.text_fc.weight.grad); put("text_fc.bias", bb.text_fc.bias.grad)
    put("tokenlearn_text.weight", torch.stack([bb.tokenlearn_text.tokenizers[s].conv[0].weight.grad.reshape(C) for s in range(8)]))
    put("tokenlearn_text.bias", torch.cat([bb.tokenlearn_text.tokenizers[s].conv[0].bias.grad for s in range(8)]))
    put("masks_text.weight", bb.masks_text.weight.grad)
    put("s_remain_map.0.weight", model.s_remain_map[0].weight.grad); put("s_remain_map.0.bias", model.s_remain_map[0].bias.grad)
    put("s_remain_map.2.weight", model.s_remain_map[2].weight.grad); put("s_remain_map.2.bias", model.s_remain_map[2].bias.grad)
    for n, p in bb.clip.named_parameters():
        if n.startswith("visual.") or n == "logit_scale":
            continue
        assert p.grad is not None, n
        put("clip." + n, p.grad)
    import json
    out["state_dict_keys"] = json.dumps({k: list(v.shape) for k, v in model.state_dict().items()
                                         if not k.startswith("backbone.clip.visual.") and not k.startswith("backbone.image_backbone.")})
    np.savez_compressed(os.path.join(OUT, "tgcir_step.npz"), **out)
    print("loss", float(loss), "entries", len(out), "bytes", os.path.getsize(os.path.join(OUT, "tgcir_step.npz")))


if __name__ == "__main__":
    main()
