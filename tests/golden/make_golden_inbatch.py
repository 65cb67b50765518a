"""Golden vectors for BASELINE config 1 (clip4cir/train.py --wo_bank): the reference's own in-batch-negative
step, `models.CIRPlus(wo_bank=True).forward` (clip4cir/models.py:151-167), on the tiny CLIP of tiny_clip.npz.

Build container only (imports /root/reference).  Stores inputs, the bbc_loss and the gradient of EVERY
parameter (visual tower included - it is trainable in this configuration, models.py:31-33).

    python tests/golden/make_golden_inbatch.py   ->  tests/golden/cirplus_inbatch.npz
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_golden import REF, install_stubs  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "clip4cir"))
    import clip      # noqa: E402
    import models    # noqa: E402

    z = np.load(os.path.join(OUT, "tiny_clip.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    tmp = "/tmp/_tiny_clip_sd_inbatch.pt"
    torch.save(sd, tmp)
    model = models.CIRPlus(tmp, tau=0.02, device=torch.device("cpu"), wo_bank=True)
    model.train()

    g = torch.Generator().manual_seed(21)
    B = 4
    tok = torch.from_numpy(z["ids"])[:B].clone()
    refer_image = torch.randn(B, 3, 32, 32, generator=g)
    target_image = torch.randn(B, 3, 32, 32, generator=g)
    clip.tokenize = lambda text, *a, **k: tok       # the tiny vocabulary has 512 rows; the tokenizer is pinned elsewhere
    model.zero_grad()
    loss = model.forward(["x"] * B, None, None, None, refer_image=refer_image, target_image=target_image)["bbc_loss"]
    loss.backward()
    out = {"ids": tok.numpy(), "refer_image": refer_image.numpy(), "target_image": target_image.numpy(),
           "tau": np.float32(0.02), "loss": loss.detach().numpy()}
    with torch.no_grad():
        out["refer_feats"] = model.clip.encode_image(refer_image).numpy()
        out["target_feats"] = model.clip.encode_image(target_image).numpy()
    for n, p in model.clip.named_parameters():
        if p.grad is not None:
            out["grad::" + n] = p.grad.numpy().copy()
        else:
            assert n == "logit_scale", n
    np.savez_compressed(os.path.join(OUT, "cirplus_inbatch.npz"), **out)
    print("loss", float(loss), "params with grad", sum(k.startswith("grad::") for k in out))


if __name__ == "__main__":
    main()
