"""Golden metrics for the fusion-style validation (tgcir/validate.py; blip4cir/validate.py is the same code with a
different fusion call): compute_fiq_val_metrics / compute_cirr_val_metrics run on CPU with a stub model whose
img_txt_fusion returns pre-made query features in call order, on a synthetic token gallery.

Build container only (imports /root/reference).   python tests/golden/make_golden_valfusion.py -> valfusion.npz
"""
import json
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_golden import REF, FakeCirrDataset, FakeFiqDataset, install_stubs  # noqa: E402


from cases import StubFusion, valfusion_inputs  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "tgcir"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    import validate                 # noqa: E402  (tgcir/validate.py)
    import torch.utils.data as tud
    real_loader = tud.DataLoader
    validate.DataLoader = lambda *a, **k: real_loader(*a, **{**k, "num_workers": 0, "pin_memory": False})
    validate.device = torch.device("cpu")

    tokens, pooled, names, q, fiq_rows, cirr_rows = valfusion_inputs()
    r10, r50 = validate.compute_fiq_val_metrics(FakeFiqDataset(fiq_rows), StubFusion(q), tokens, pooled, names)
    cirr = validate.compute_cirr_val_metrics(FakeCirrDataset(cirr_rows), StubFusion(q), tokens, pooled, names)
    np.savez_compressed(os.path.join(OUT, "valfusion.npz"), fiq=np.array([r10, r50]), cirr=np.array(cirr))
    print("fiq", r10, r50, "cirr", cirr)


if __name__ == "__main__":
    main()
