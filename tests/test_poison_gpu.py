"""Uninitialised-scratch screen: with SPN_DEBUG_POISON=1 every scratch allocation of the Python layer (activation arenas,
backward workspaces, the bank pair's partial buffer, the op workspaces) is filled with 0xFF bytes - NaN as fp32 and as bf16.
A kernel that reads scratch it never wrote then turns its output into NaN instead of into allocator-dependent noise.  The
training steps of the CLIP text path (dense and packed) and of the BLIP fusion path must stay finite."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from spn4cir_amd import ops, synthetic
assert ops._POISON
from spn4cir_amd.models import CIRPlus
from spn4cir_amd.trainer import Stage2Trainer
dev = torch.device("cuda", 0)
# CLIP text path: 2 layers x 128, B = 24, M = 3 000 (fused single-pass bank step), dense then packed, and a 100-row fp8 bank
W, layers, D, vocab = 128, 2, 128, 1000
from oracle import clip_text
sd = clip_text.synthetic_text_state_dict(W, layers, D, vocab=vocab, seed=0)
ids = clip_text.synthetic_token_ids(24, vocab=vocab, seed=1)
target, refer = synthetic.banks(3000, D, seed=2)
ridx, labels = synthetic.triplet_indices(24, 3000, seed=4)
for bank_dtype in ("bf16", "fp8"):
    model = CIRPlus(sd, tau=0.02, device=dev, plus=True)
    tr = Stage2Trainer(model, lr=1e-4)
    tr.set_banks(refer, target, bank_dtype=bank_dtype)
    for packed in (False, True):
        if packed:
            cu, total = tr.tower.cu_seqlens(ids)
            loss = tr.step(ids.to(dev), ridx.to(dev), labels.to(dev), cu.to(dev), total)
        else:
            loss = tr.step(ids.to(dev), ridx.to(dev), labels.to(dev))
        assert torch.isfinite(loss).all(), (bank_dtype, packed, loss)
        assert torch.isfinite(tr.tower.grads).all(), (bank_dtype, packed, "grads")
        assert torch.isfinite(tr.tower.params).all(), (bank_dtype, packed, "params")
# BLIP fusion path on the golden 2-layer encoder
import test_fusion_gpu as T
from spn4cir_amd.fusion import BlipStage2Trainer
z, enc = T._blip_setup(os.path.join(os.getcwd(), "tests", "golden"))
bt = BlipStage2Trainer(enc, tau=float(z["tau"]), lr=1e-3)
bt.set_bank(torch.from_numpy(z["bank"]))
for _ in range(2):
    loss = bt.step(torch.from_numpy(z["ids"]).cuda(), torch.from_numpy(z["mask"]).cuda(), torch.from_numpy(z["enc"]).cuda(),
                   torch.from_numpy(z["labels"]).cuda())
    assert torch.isfinite(loss).all() and torch.isfinite(enc.grads).all() and torch.isfinite(enc.params).all()
print("POISON_OK")
"""


@pytest.mark.gpu
def test_training_steps_read_no_uninitialised_scratch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, "-c", _CHILD], env=dict(os.environ, SPN_DEBUG_POISON="1"), cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert "POISON_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
