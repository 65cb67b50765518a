"""Golden vectors for CLIP's ModifiedResNet image tower: a tiny `CLIP` with vision_layers given as a tuple (which
selects ModifiedResNet, clip/model.py:263-271), randomised BatchNorm statistics, eval mode -> encode_image output.
Build container only.   python tests/golden/make_golden_resnet.py  ->  tests/golden/tiny_clip_resnet.npz
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
    from clip.model import CLIP  # noqa: E402
    torch.manual_seed(0)
    m = CLIP(128, 64, (1, 2, 1, 1), 8, None, 77, 64, 64, 1, 1).eval()       # embed 128, res 64, RN layers, width 8
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for n, p in m.visual.named_parameters():
            if ".bn" in n or "downsample.1" in n or n.startswith("bn"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g) if n.endswith("weight") else 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        for n, b in m.visual.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif n.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    image = torch.randn(3, 3, 64, 64, generator=g)
    with torch.no_grad():
        feats = m.encode_image(image)
    out = {"sd::" + k: v.detach().numpy() for k, v in m.state_dict().items() if k.startswith("visual.")}
    out.update(image=image.numpy(), image_feats=feats.numpy())
    np.savez_compressed(os.path.join(OUT, "tiny_clip_resnet.npz"), **out)
    print("feats", feats.shape, float(feats.abs().max()))


if __name__ == "__main__":
    main()
