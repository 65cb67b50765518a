"""Negative-type ablation losses (clip4cir/models_negtype.py) on the HIP path: the feature-level head against vectors captured
from the reference's own loss methods (tests/golden/negtype.npz) and against the oracle at a larger shape; the end-to-end
`CIRPlus(neg_type=7).forward` against the reference's forward + autograd on the tiny CLIP (every parameter gradient)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_negtype_head_matches_reference_capture(golden_dir, tag):
    """fp32 head: loss to 1e-5 relative, feature gradients to 1e-4 relative L2, for every captured neg_type mask."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spn4cir_amd import ops
    z = np.load(os.path.join(golden_dir, "negtype.npz"))
    r, t, i = (torch.from_numpy(z[f"{tag}::{k}"]).cuda() for k in ("refer", "text", "target"))
    tau = float(z[f"{tag}::tau"])
    for nt in (1, 2, 4, 8, 7, 15, 5, 10):
        loss, dr, dt, di = ops.negtype_head(r, t, i, tau, nt)
        ref = float(z[f"{tag}::{nt}::loss"])
        assert abs(loss.item() - ref) < 1e-5 * max(1.0, abs(ref)), (nt, loss.item(), ref)
        for g, k in ((dr, "refer"), (dt, "text"), (di, "target")):
            assert _rel(g, torch.from_numpy(z[f"{tag}::{nt}::d_{k}"])) < 1e-4, (nt, k)
        again = ops.negtype_head(r, t, i, tau, nt)
        assert all(torch.equal(x, y) for x, y in zip((loss, dr, dt, di), again))      # fixed summation order


@pytest.mark.parametrize("B,D,tau,nt", [(64, 512, 0.01, 15), (33, 768, 0.02, 3), (128, 1024, 0.01, 7), (1, 64, 0.02, 15)])
def test_negtype_head_matches_oracle(B, D, tau, nt):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import negtype
    from spn4cir_amd import ops
    g = torch.Generator().manual_seed(B + D + nt)
    r, t = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g) * 0.7
    i = torch.randn(B, D, generator=g) + 0.1 * (r + t)
    rr, tt, ii = (x.double().clone().requires_grad_(True) for x in (r, t, i))
    ref = negtype.loss(rr, tt, ii, tau, nt)
    ref.backward()
    loss, dr, dt, di = ops.negtype_head(r.cuda(), t.cuda(), i.cuda(), tau, nt)
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    for got, want in ((dr, rr.grad), (dt, tt.grad), (di, ii.grad)):
        assert _rel(got, want) < 2e-4


def test_negtype_model_matches_reference_forward(golden_dir):
    """models_negtype.CIRPlus(neg_type=7).forward on the tiny CLIP: loss within 1e-2, every parameter gradient (text + visual)
    within 5e-2 relative L2 of the reference's forward + autograd (bf16 towers against its fp32 CPU path)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spn4cir_amd.models_negtype import CIRPlus
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    s = np.load(os.path.join(golden_dir, "negtype.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    model = CIRPlus(sd, tau=0.02, device=torch.device("cuda"), neg_type=7)
    out = model.forward(torch.from_numpy(s["e2e::ids"]), None, None, None, refer_image=torch.from_numpy(s["e2e::refer_image"]).cuda(),
                        target_image=torch.from_numpy(s["e2e::target_image"]).cuda())
    loss = out["bbc_loss"]
    ref = float(s["e2e::loss"])
    assert abs(loss.item() - ref) < 1e-2 * max(1.0, abs(ref)), (loss.item(), ref)
    loss.backward()
    named = dict(model.clip.named_parameters())
    worst, n = (0.0, None), 0
    for k in s.files:
        if not k.startswith("e2e::grad::"):
            continue
        name = k[len("e2e::grad::"):]
        g, r = named[name].grad, torch.from_numpy(s[k])
        assert g is not None, name
        err = _rel(g, r)
        worst = max(worst, (err, name))
        assert err < 5e-2, (name, err)
        n += 1
    assert n == 61
    print("negtype e2e: loss", loss.item(), "ref", ref, "worst gradient error", worst)
