"""Golden vectors for the negative-type ablation losses (clip4cir/models_negtype.py:53-134): the reference's own
text_neg_loss / refer_neg_loss / infonce_loss and forward()'s neg_type mask, evaluated at the FEATURE level on seeded
inputs (the towers are pinned by tiny_clip.npz / cirplus_inbatch.npz), plus one end-to-end forward on the tiny CLIP.

Build container only (imports /root/reference).

    python tests/golden/make_golden_negtype.py   ->  tests/golden/negtype.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_golden import REF, install_stubs  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "clip4cir"))
    import clip             # noqa: E402
    import models_negtype   # noqa: E402

    z = np.load(os.path.join(OUT, "tiny_clip.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    tmp = "/tmp/_tiny_clip_sd_negtype.pt"
    torch.save(sd, tmp)
    out = {}
    g = torch.Generator().manual_seed(33)
    for tag, (B, D, tau) in {"a": (6, 64, 0.02), "b": (17, 128, 0.01)}.items():
        refer = torch.randn(B, D, generator=g)
        text = torch.randn(B, D, generator=g) * 0.7
        target = torch.randn(B, D, generator=g) + 0.12 * (refer + text)       # weakly correlated: neither uniform nor saturated softmaxes
        out[f"{tag}::refer"], out[f"{tag}::text"], out[f"{tag}::target"] = refer.numpy(), text.numpy(), target.numpy()
        out[f"{tag}::tau"] = np.float32(tau)
        for nt in (1, 2, 4, 8, 7, 15, 5, 10):
            model = models_negtype.CIRPlus(tmp, tau=tau, device=torch.device("cpu"), neg_type=nt)
            r, t, i = (x.clone().requires_grad_(True) for x in (refer, text, target))
            tn = F.normalize(i)                                                   # forward(): :101
            qn = F.normalize(model.combining_function(r, t))                      # :102-103
            tl = model.infonce_loss(qn, tn, tau=model.tau)                        # :104
            rl = model.refer_neg_loss(r, t, tn)                                   # :105
            xl = model.text_neg_loss(r, t, tn)                                    # :106
            ql = model.infonce_loss(tn, qn, tau=model.tau)                        # :107
            loss, cnt, k = torch.tensor(0.0), 0, nt                               # :108-127
            if k // 8 == 1:
                loss, cnt = loss + ql, cnt + 1
            k %= 8
            if k // 4 == 1:
                loss, cnt = loss + tl, cnt + 1
            k %= 4
            if k // 2 == 1:
                loss, cnt = loss + xl, cnt + 1
            k %= 2
            if k == 1:
                loss, cnt = loss + rl, cnt + 1
            loss = loss / cnt
            loss.backward()
            out[f"{tag}::{nt}::loss"] = loss.detach().numpy()
            out[f"{tag}::{nt}::d_refer"], out[f"{tag}::{nt}::d_text"], out[f"{tag}::{nt}::d_target"] = \
                r.grad.numpy().copy(), t.grad.numpy().copy(), i.grad.numpy().copy()
    # end to end on the tiny CLIP: models_negtype.CIRPlus.forward itself (neg_type 7 = the default-ish mix of three terms)
    tok = torch.from_numpy(z["ids"])[:4].clone()
    clip.tokenize = lambda text, *a, **k: tok
    g2 = torch.Generator().manual_seed(21)
    refer_image, target_image = torch.randn(4, 3, 32, 32, generator=g2), torch.randn(4, 3, 32, 32, generator=g2)
    model = models_negtype.CIRPlus(tmp, tau=0.02, device=torch.device("cpu"), neg_type=7)
    model.train()
    model.zero_grad()
    loss = model.forward(["x"] * 4, None, None, None, refer_image=refer_image, target_image=target_image)["bbc_loss"]
    loss.backward()
    out["e2e::ids"], out["e2e::refer_image"], out["e2e::target_image"] = tok.numpy(), refer_image.numpy(), target_image.numpy()
    out["e2e::loss"] = loss.detach().numpy()
    for n, p in model.clip.named_parameters():
        if p.grad is not None:
            out["e2e::grad::" + n] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "negtype.npz"), **out)
    print("e2e loss", float(loss), {k: float(v) for k, v in out.items() if k.endswith("::loss") and k.startswith("a::")})


if __name__ == "__main__":
    main()
