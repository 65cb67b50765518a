"""oracle/noise.py (the bf16 operand-rounding model behind the GPU gradient gates) and the committed floor file."""
import json
import os

import torch

from oracle import noise


def _rb(x):
    return x.to(torch.bfloat16).float()


def test_products_run_on_rounded_operands_forward_and_backward():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(5, 7, 16, generator=g, requires_grad=True)
    w = torch.randn(12, 16, generator=g, requires_grad=True)
    go = torch.randn(5, 7, 12, generator=g)
    with noise.bf16_gemm_operands() as m:
        y = a @ w.t()
        y.backward(go)
    assert m.products == 1
    ar, wr, gr = _rb(a.detach()), _rb(w.detach()), _rb(go)
    assert torch.equal(y.detach(), ar @ wr.t())                                   # forward: rounded operands, fp32 accumulate, fp32 out
    assert torch.allclose(a.grad, gr @ wr, atol=1e-6, rtol=1e-6)                  # dA = R(g) R(W)
    assert torch.allclose(w.grad, (gr.reshape(-1, 12).t() @ ar.reshape(-1, 16)), atol=1e-5, rtol=1e-5)   # dW = R(g)^T R(A)
    # fp64 products (the oracle's reference scoring) are left alone; outputs=True also rounds the result
    b = torch.randn(4, 4, generator=g, dtype=torch.float64)
    with noise.bf16_gemm_operands(outputs=True) as m2:
        assert torch.equal(b @ b, torch.matmul(b, b)) and m2.products == 0
        z = torch.nn.functional.linear(a.detach(), w.detach())
    assert torch.equal(z, _rb(ar @ wr.t()))
    # product + bias and the residual form x + product + bias: the bias gets the column sums of the ROUNDED output gradient, the
    # residual branch the unrounded one
    b = torch.randn(12, generator=g, requires_grad=True)
    x = torch.randn(5, 7, 12, generator=g, requires_grad=True)
    a2, w2 = a.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    with noise.bf16_gemm_operands():
        (x + a2 @ w2.t() + b).backward(go)
    assert torch.allclose(b.grad, gr.reshape(-1, 12).sum(0), atol=1e-6) and torch.equal(x.grad, go)
    assert torch.allclose(a2.grad, gr @ wr, atol=1e-6, rtol=1e-6)
    # outside the context nothing is rounded
    assert torch.equal(a.detach() @ w.detach().t(), torch.matmul(a.detach(), w.detach().t()))


def test_committed_floor_file_is_complete_and_reproducible(golden_dir):
    with open(os.path.join(golden_dir, "noise_floor.json")) as f:
        data = json.load(f)
    want = {"vitl14_b8": 149, "vitl14_b8_e4m3": 149, "config1_vitb32_b4": 301, "blip_768": 318, "blip_1024": 318,
            "blip_768_refinit": 318, "blip_1024_refinit": 318}
    for case, n in want.items():
        assert len(data[case]["operands"]) == n and len(data[case]["autocast"]) == n and data[case]["realisations"] == 4, case
        assert all(data[case]["operands_max"][k] >= v * 0.999 for k, v in data[case]["operands"].items()), case   # RMS <= largest sample
        vals = [v for k, v in data[case]["operands"].items() if not k.endswith("self.key.bias")]
        assert all(0 < v < 0.12 for v in vals), case                         # bf16 operand noise: 0.3-10 % per tensor, never "free"
        assert data[case]["operands_feat_max_1_minus_cos"] < 1e-3              # the north_star feature gate is far above the floor
    # regenerate the cheapest case from the committed script: same floors (thread count may move the last bits of a sum)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_noise_floor", os.path.join(golden_dir, "make_noise_floor.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    again = mod.case_config1()["operands"]
    ref = data["config1_vitb32_b4"]["operands"]
    assert set(again) == set(ref)
    worst = max(abs(again[k] - ref[k]) / ref[k] for k in ref)
    assert worst < 0.05, worst
