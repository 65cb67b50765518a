"""GPU parity tests: every HIP kernel, called through the C-ABI, against the CPU oracle.

Tolerances (stated per test): GEMM-type kernels take bf16 inputs and accumulate in fp32, so
they are compared with an fp64 reference evaluated on the SAME bf16-rounded inputs; bf16
outputs add one rounding (2^-9 relative)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spn4cir_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (308, 192, 128), (462, 64, 256), (1000, 2304, 768), (77, 512, 3072),
                                   (2050, 2304, 192)])
def test_gemm_nt_plain_and_bias(ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(N, K, generator=g))
    bias = torch.randn(N, generator=g)
    ref = a.double() @ b.double().t() + bias.double()
    out32 = ops.gemm_nt(dev(a), dev(b), dev(bias), out_dtype=torch.float32)
    assert rel_err(out32, ref) < 2e-5          # fp32 accumulation of exact bf16 products
    out16 = ops.gemm_nt(dev(a), dev(b), dev(bias))
    assert rel_err(out16, ref) < 6e-3          # + one bf16 rounding of the output


def test_gemm_nt_is_not_transposed(ops):
    # A = I-like / asymmetric B catches a swapped C layout (cdna guide rule 16)
    M = N = 128
    K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    b = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 / 16.0
    ref = a.double() @ bf(b).double().t()
    out = ops.gemm_nt(dev(bf(a)), dev(bf(b)), out_dtype=torch.float32)
    assert torch.equal(out.cpu().double(), ref)


@pytest.mark.parametrize("M,N", [(300, 256),        # fewer 128x128 tiles than CUs: the 64-row tiles of the 128-wide kernel
                                 (2100, 2048)])     # 272 tiles of 128x128, 72 of 256x256: its 128-row tiles
def test_gemm_nt_epilogues(ops, M, N):
    g = torch.Generator().manual_seed(5)
    K = 128
    a, b = bf(torch.randn(M, K, generator=g) * 0.5), bf(torch.randn(N, K, generator=g) * 0.2)
    bias = torch.randn(N, generator=g) * 0.1
    resid = torch.randn(M, N, generator=g)
    lin = a.double() @ b.double().t() + bias.double()
    # quick-gelu with the pre-activation copy
    u, pre = ops.gemm_nt(dev(a), dev(b), dev(bias), act=ops.ACT_QUICKGELU, want_pre=True)
    assert rel_err(pre, lin) < 6e-3
    assert rel_err(u, quick_gelu(lin)) < 8e-3
    # exact gelu
    u2 = ops.gemm_nt(dev(a), dev(b), dev(bias), act=ops.ACT_GELU_ERF, out_dtype=torch.float32)
    assert rel_err(u2, torch.nn.functional.gelu(lin)) < 1e-4
    # residual
    r = ops.gemm_nt_resid(dev(a), dev(b), dev(bias), dev(resid))
    assert rel_err(r, lin + resid.double()) < 2e-5
    # d-activation: (a b^T) * act'(pre)
    pre_b = bf(torch.randn(M, N, generator=g))
    x = pre_b.double().requires_grad_(True)
    quick_gelu(x).sum().backward()
    d = ops.gemm_nt_dact(dev(a), dev(b), dev(pre_b), ops.ACT_QUICKGELU)
    assert rel_err(d, (a.double() @ b.double().t()) * x.grad) < 8e-3
    x2 = pre_b.double().requires_grad_(True)
    torch.nn.functional.gelu(x2).sum().backward()
    d2 = ops.gemm_nt_dact(dev(a), dev(b), dev(pre_b), ops.ACT_GELU_ERF)
    assert rel_err(d2, (a.double() @ b.double().t()) * x2.grad) < 8e-3


@pytest.mark.parametrize("M,N,K", [(4352, 4096, 128), (4352, 4096, 192), (19712, 3072, 768), (8960, 2304, 768), (19712, 768, 768)])
def test_gemm_nt_persistent_matches_one_tile_kernel(ops, M, N, K):
    """gemm_nt2p (one workgroup per CU walking several tiles, operand stream running across tile boundaries) against the
    one-tile-per-workgroup kernel it replaces on multi-round products: every epilogue BIT-identical (same MFMA order per output,
    same epilogue arithmetic), odd and even k-tile counts, a ragged last round (272 = 256 + 16, 924 = 3 x 256 + 156 tiles), and
    a single-round shape that must not take it.  Plus a race screen: 12 launches under a bandwidth-heavy side stream."""
    from spn4cir_amd import _lib
    L = _lib.lib()
    if not _lib.config_dump()["experiments_build"]:
        assert L.spn_gemm_config(0, 1) == -1               # the shipped library refuses the kernel it does not contain
        pytest.skip("gemm_nt2p is compiled only into the experiments build (tests/test_experiments_gpu.py runs this there)")
    g = torch.Generator().manual_seed(M + N + K)
    a, b = dev(bf(torch.randn(M, K, generator=g) * 0.5)), dev(bf(torch.randn(N, K, generator=g) * 0.2))
    bias = dev(torch.randn(N, generator=g) * 0.1)
    pre_b = dev(bf(torch.randn(M, N, generator=g)))
    resid = dev(torch.randn(M, N, generator=g))

    def run():
        u, pre = ops.gemm_nt(a, b, bias, act=ops.ACT_QUICKGELU, want_pre=True)
        return [ops.gemm_nt(a, b, bias), ops.gemm_nt(a, b, bias, out_dtype=torch.float32), u, pre,
                ops.gemm_nt_dact(a, b, pre_b, ops.ACT_QUICKGELU), ops.gemm_nt_dact(a, b, pre_b, ops.ACT_GELU_ERF),
                ops.gemm_nt_resid(a, b, bias, resid)]
    try:
        assert L.spn_gemm_config(0, 0) == 0
        want = [t.clone() for t in run()]
        assert L.spn_gemm_config(0, 1) == 0
        got = run()
        for i, (x, y) in enumerate(zip(got, want)):
            assert torch.equal(x, y), (i, (x.float() - y.float()).abs().max().item())
        ref = a.double() @ b.double().t() + bias.double()
        assert rel_err(got[1], ref) < 2e-5
        side = torch.cuda.Stream()
        big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        for it in range(12):
            if it % 2:
                with torch.cuda.stream(side):
                    big.add_(1.0)
            assert torch.equal(ops.gemm_nt(a, b, bias), want[0]), it
        torch.cuda.synchronize()
    finally:
        L.spn_gemm_config(0, -1)


@pytest.mark.parametrize("Kr,N1,N2", [(64, 128, 128), (462, 128, 384), (1000, 64, 512), (19712, 768, 768), (77 * 6, 384, 128),
                                      (6, 128, 64)])
def test_gemm_tn(ops, Kr, N1, N2):
    g = torch.Generator().manual_seed(Kr + N1)
    a, b = bf(torch.randn(Kr, N1, generator=g)), bf(torch.randn(Kr, N2, generator=g))
    ref = a.double().t() @ b.double()
    out, cs = ops.gemm_tn(dev(a), dev(b), want_colsum=True)
    assert rel_err(out, ref) < 3e-5
    assert rel_err(cs, a.double().sum(0)) < 3e-5      # fused bias gradient (column sums of A)
    # alpha + accumulate path
    base = torch.randn(N1, N2, generator=g)
    acc = dev(base.clone())
    ops.gemm_tn(dev(a), dev(b), alpha=0.5, out=acc, accumulate=True)
    assert rel_err(acc, base.double() + 0.5 * ref) < 3e-5


@pytest.mark.parametrize("Kr,sa,sb", [(19712, (2304, 768), (768, 768)), (1000, (512, 256), (256, 264)),
                                      (70, (256, 256), (8, 520)), (4099, (768, 3072), (128, 64))])
def test_gemm_tn_pair(ops, Kr, sa, sb):
    """Two weight-gradient products in one launch (tiles of the second follow the first's; shared split-K): each
    must equal its own fp64 product, ragged tiles and split boundaries included."""
    g = torch.Generator().manual_seed(Kr)
    a1, b1 = bf(torch.randn(Kr, sa[0], generator=g)), bf(torch.randn(Kr, sa[1], generator=g))
    a2, b2 = bf(torch.randn(Kr, sb[0], generator=g)), bf(torch.randn(Kr, sb[1], generator=g))
    c1, s1, c2, s2 = ops.gemm_tn_pair(dev(a1), dev(b1), dev(a2), dev(b2))
    assert rel_err(c1, a1.double().t() @ b1.double()) < 3e-5
    assert rel_err(c2, a2.double().t() @ b2.double()) < 3e-5
    assert rel_err(s1, a1.double().sum(0)) < 3e-5 and rel_err(s2, a2.double().sum(0)) < 3e-5
    # canary: same inputs again give the same bits (no cross-problem race on the shared workspace)
    d1, t1, d2, t2 = ops.gemm_tn_pair(dev(a1), dev(b1), dev(a2), dev(b2))
    assert torch.equal(c1, d1) and torch.equal(c2, d2) and torch.equal(s1, t1) and torch.equal(s2, t2)


def test_gemm_tn_asymmetric(ops):
    # one-hot A picks single rows of B: exact, catches any row/column permutation in the
    # transpose-read fragments
    Kr, N1, N2 = 128, 128, 128
    a = torch.zeros(Kr, N1)
    perm = (torch.arange(N1) * 37 + 11) % Kr
    a[perm, torch.arange(N1)] = 1.0                      # column n1 has its 1 at row perm[n1]
    b = (torch.arange(Kr * N2, dtype=torch.float32).reshape(Kr, N2) % 509) / 8.0
    out = ops.gemm_tn(dev(bf(a)), dev(bf(b)))
    assert torch.equal(out.cpu(), bf(b).float()[perm])


def test_cast_transpose_colsum(ops):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(300, 200, generator=g)
    y, yt = ops.cast_transpose_bf16(dev(x))
    assert torch.equal(y.cpu(), bf(x)) and torch.equal(yt.cpu(), bf(x).t())
    assert torch.equal(ops.cast_bf16(dev(x)).cpu(), bf(x))
    xb = bf(torch.randn(1234, 384, generator=g))
    assert rel_err(ops.colsum(dev(xb)), xb.double().sum(0)) < 1e-5


# ------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,W", [(7, 128), (462, 768), (50, 512), (33, 1024)])
def test_layernorm(ops, rows, W):
    g = torch.Generator().manual_seed(rows + W)
    x = torch.randn(rows, W, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(W, generator=g), torch.randn(W, generator=g)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yref = torch.nn.functional.layer_norm(xd, (W,), gd, bd, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), out_dtype=torch.float32)
    assert rel_err(y, yref.detach()) < 1e-5
    yb, _, _ = ops.layernorm_fwd(dev(x), dev(gamma), dev(beta))
    assert rel_err(yb, yref.detach()) < 6e-3
    dy = torch.randn(rows, W, generator=g)
    yref.backward(dy.double())
    dx, dxb, dg, db = ops.layernorm_bwd(dev(dy), dev(x), dev(gamma), mean, rstd)
    assert rel_err(dx, xd.grad) < 2e-5 and rel_err(dg, gd.grad) < 2e-5 and rel_err(db, bd.grad) < 2e-5
    assert rel_err(dxb, xd.grad) < 6e-3
    # bf16 dy + accumulate into an existing dx
    base = torch.randn(rows, W, generator=g)
    dyb = bf(dy)
    xd2 = x.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd2, (W,), gamma.double(), beta.double(), 1e-5).backward(dyb.double())
    acc = dev(base.clone())
    ops.layernorm_bwd(dev(dyb), dev(x), dev(gamma), mean, rstd, dx_accum=acc)
    assert rel_err(acc, base.double() + xd2.grad) < 2e-5


# ------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, B, H, Lq, Lk, causal, key_bias, scale):
    q = q.double().view(B, Lq, H, 64).transpose(1, 2)
    k = k.double().view(B, Lk, H, 64).transpose(1, 2)
    v = v.double().view(B, Lk, H, 64).transpose(1, 2)
    s = q @ k.transpose(-1, -2) * scale
    if key_bias is not None:
        s = s + key_bias.double()[:, None, None, :]
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf"), dtype=torch.float64).triu_(1)
    p = torch.softmax(s, -1)
    o = (p @ v).transpose(1, 2).reshape(B * Lq, H * 64)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,H,Lq,Lk,causal,bias", [(3, 2, 77, 77, True, False), (2, 3, 50, 50, False, False),
                                                   (1, 2, 257, 257, False, False), (2, 2, 9, 577, False, False),
                                                   (3, 2, 20, 20, False, True), (2, 1, 130, 130, True, False),
                                                   (2, 2, 32, 577, False, True), (1, 3, 64, 200, False, False),
                                                   (2, 1, 40, 100, False, True), (2, 2, 77, 77, True, True),
                                                   # every wave count of the whole-head kernels, tile-edge lengths
                                                   (2, 2, 1, 1, True, False), (2, 2, 16, 16, False, False),
                                                   (2, 2, 17, 17, True, False), (1, 2, 48, 48, True, True),
                                                   (2, 1, 64, 64, False, False), (1, 1, 65, 65, True, False),
                                                   (1, 2, 96, 96, False, True), (1, 2, 113, 113, False, True),
                                                   (1, 1, 128, 128, True, False), (2, 2, 16, 577, False, False),
                                                   (1, 2, 64, 65, False, True), (1, 1, 33, 100, False, False),
                                                   # more (sequence, head) pairs than resident workgroups: the persistent backward
                                                   # walk (two pairs per workgroup) loops; 513 pairs leave the last slot idle
                                                   (70, 12, 77, 77, True, False), (171, 3, 40, 40, False, True)])
def test_attention_fwd_bwd(ops, B, H, Lq, Lk, causal, bias):
    g = torch.Generator().manual_seed(B * 1000 + Lq)
    W = H * 64
    if Lq == Lk:     # packed qkv, as in the towers
        qkv = bf(torch.randn(B * Lq, 3 * W, generator=g))
        q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
        dq_, dk_, dv_ = dev(qkv)[:, :W], dev(qkv)[:, W:2 * W], dev(qkv)[:, 2 * W:]
    else:
        q, k, v = (bf(torch.randn(B * Lq, W, generator=g)), bf(torch.randn(B * Lk, W, generator=g)),
                   bf(torch.randn(B * Lk, W, generator=g)))
        dq_, dk_, dv_ = dev(q), dev(k), dev(v)
    kb = None
    if bias:
        kb = torch.zeros(B, Lk)
        for b in range(B):
            kb[b, Lk - 1 - 3 * b:] = -10000.0        # BERT-style padding mask (blip4cir/med.py:686)
    qd, kd, vd = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    oref, lseref = attn_ref(qd, kd, vd, B, H, Lq, Lk, causal, kb, 0.125)
    o, lse = ops.attention_fwd(dq_, dk_, dv_, B, H, Lq, Lk, causal=causal, key_bias=None if kb is None else dev(kb))
    assert rel_err(o, oref.detach()) < 1e-2          # P and O are rounded to bf16
    assert (lse.cpu().double() - lseref.detach()).abs().max() < 2e-3
    d_o = bf(torch.randn(B * Lq, W, generator=g))
    oref.backward(d_o.double())
    dq, dk, dv = ops.attention_bwd(dq_, dk_, dv_, o, lse, dev(d_o), B, H, Lq, Lk, causal=causal,
                                   key_bias=None if kb is None else dev(kb))
    assert rel_err(dq, qd.grad) < 2e-2 and rel_err(dk, kd.grad) < 2e-2 and rel_err(dv, vd.grad) < 2e-2


# ------------------------------------------------------------------------------- bank loss
def _bank_case(B, M, D, seed):
    g = torch.Generator().manual_seed(seed)
    text = torch.randn(B, D, generator=g)
    refer = torch.randn(M, D, generator=g)
    bank = torch.nn.functional.normalize(torch.randn(M, D, generator=g))
    ridx = torch.randint(0, M, (B,), generator=g)
    labels = torch.randint(0, M, (B,), generator=g)
    labels[0] = 0
    labels[-1] = M - 1
    return text, refer, bank, ridx, labels


@pytest.mark.parametrize("B,M,D,tau", [(4, 500, 64, 0.01), (32, 4099, 512, 0.02), (16, 40000, 768, 0.03),
                                        (70, 3000, 768, 0.02), (256, 40000, 768, 0.02), (33, 1500, 640, 0.02),
                                        (2048, 5000, 768, 0.02), (512, 20000, 768, 0.02),
                                        (1, 7, 128, 0.02), (3, 33, 256, 0.01)])   # bank smaller than one tile, B = 1
def test_bank_infonce(ops, B, M, D, tau):
    from oracle import bank_loss
    text, refer, bank, ridx, labels = _bank_case(B, M, D, B + M)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    qref = bank_loss.l2_normalize(refer[ridx] + text)
    assert rel_err(q, qref) < 1e-6
    Dp = ops.bank_dim(D)
    assert qb.shape == (B, Dp) and torch.equal(qb[:, :D].cpu(), bf(q.cpu())) and not qb[:, D:].any()
    bank_b = ops.prepare_bank(dev(bank))
    # reference on the same bf16-rounded operands: only accumulation order differs
    qr, br = qb[:, :D].cpu().float(), bank_b[:, :D].cpu().float()
    lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(qr, br, labels, tau)
    stats = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    assert (lse.cpu().double() - lse_ref).abs().max() < 2e-4
    assert (stats[:, 3].cpu().double() - lab_ref).abs().max() < 2e-4
    assert abs(mean.item() - row_ref.mean().item()) < 2e-4
    # the bf16 operand rounding itself: loss within 1e-2 of the fp32 oracle (tau amplifies x50..100)
    full = bank_loss.infonce(qref, bank, labels, tau).item()
    assert abs(mean.item() - full) < 1e-2 * max(1.0, abs(full))
    # gradient
    gs = 1.0 / B
    dq = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, gs)
    dq_ref = bank_loss.infonce_grad_q(qr, br, labels, tau)
    assert rel_err(dq[:, :D], dq_ref) < 1.5e-2     # G is rounded to bf16 before the second GEMM
    assert not dq[:, D:].any()
    # combiner backward
    dtext = ops.combine_l2norm_bwd(q, inv, dq[:, :D].contiguous())
    x = (refer[ridx] + text).double().requires_grad_(True)
    (bank_loss.l2_normalize(x) * dq[:, :D].cpu().double()).sum().backward()
    assert rel_err(dtext, x.grad) < 1e-5


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "e4m3"])
@pytest.mark.parametrize("B,M,D,tau", [(32, 40000, 768, 0.02), (4, 500, 128, 0.01), (32, 4099, 512, 0.02), (70, 3000, 768, 0.02),
                                        (33, 1500, 640, 0.02), (1, 7, 128, 0.02), (3, 33, 256, 0.01), (127, 9001, 1024, 0.05),
                                        (16, 100000, 768, 0.02)])
def test_bank_saved_logits_pair(ops, B, M, D, tau, fp8):
    """The forward/backward pair that keeps the step's logits (spn_bank_stats_fwd_save / spn_bank_grad_q_saved, batches
    below 128 queries: csrc/bank2.hip): statistics and loss as the oracle on the same (bf16-rounded resp. dequantised)
    operands, the saved logits themselves, dq against the oracle and against the recomputing backward kernel, two
    unequal shards with label smoothing, a bank smaller than one tile, rows that end in the middle of a 16-row tile."""
    from oracle import bank_loss
    from spn4cir_amd import _lib
    if not _lib.config_dump()["experiments_build"]:
        assert _lib.lib().spn_bank_config(1) == -1           # the shipped library refuses the mode it does not contain
        pytest.skip("bank2.hip is compiled only into the experiments build (tests/test_experiments_gpu.py runs this there)")
    _lib.lib().spn_bank_config(1)                  # the streaming pair is opt-in (spn_bank_config / SPN_BANK2=1)
    try:
        _saved_pair_case(ops, bank_loss, B, M, D, tau, fp8)
    finally:
        _lib.lib().spn_bank_config(0)


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "e4m3"])
@pytest.mark.parametrize("B,M,D,tau", [(32, 40000, 768, 0.02), (4, 500, 128, 0.01), (32, 4099, 512, 0.02), (70, 3000, 768, 0.02),
                                        (33, 1500, 640, 0.02), (1, 7, 128, 0.02), (3, 33, 256, 0.01), (127, 9001, 1024, 0.05),
                                        (16, 100000, 768, 0.02), (129, 2000, 256, 0.02), (200, 6000, 768, 0.01),
                                        (4, 540000, 128, 0.02)])      # > 2 048 rows per chunk: the e4m3 kernel with a bf16 tile image
def test_bank_fused_single_pass(ops, B, M, D, tau, fp8):
    """Default routing of the pair: ONE pass over the bank computes the softmax statistics and the unnormalised query
    gradient (flash-attention recurrence: running row maximum + rescale), the backward call only folds the chunk
    partials with the global lse and subtracts the label row.  Same checks as the two-pass pair (the save buffer holds
    the chunk partials, not logits); with label smoothing the backward call recomputes."""
    from oracle import bank_loss
    if fp8 and B >= 256:
        pytest.skip("e4m3 banks at B >= 256 run the two-pass path (bank expanded once per pass)")
    _saved_pair_case(ops, bank_loss, B, M, D, tau, fp8, check_z=False)


def test_bank_fp8_large_batch_uses_kept_image(ops):
    """From ops.FP8_IMAGE_MIN_B queries per call on an e4m3 bank is scored through its bf16 image, expanded once and kept
    (spn_bank_dequant_fp8): the image is bf16(e4m3 x row scale) bit for bit, the pair of calls gives exactly what the bf16
    entry points give on that image (saved-probabilities backward included), and both match the oracle on the dequantised
    bank."""
    from oracle import bank_loss
    B, M, D, tau = 256, 5000, 768, 0.02
    text, refer, bank, ridx, labels = _bank_case(B, M, D, 11)
    fb = ops.prepare_bank(dev(bank), "fp8")
    img = fb.bf16_image()
    assert img is fb.bf16_image()                                  # kept
    assert torch.equal(img.cpu(), fb.dequantize().to(torch.bfloat16).cpu())
    assert B >= ops.FP8_IMAGE_MIN_B
    _, qb, _ = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    out = []
    for bk in (fb, img):
        save = ops.bank_logits_buffer(B, M, "cuda")
        stats = ops.bank_stats_fwd(qb, bk, dev(labels), 1.0 / tau, save=save)
        lse, row, mean = ops.bank_loss_finalize(stats, M)
        dq = ops.bank_grad_q(qb, bk, dev(labels), 1.0 / tau, lse, 1.0 / B, saved=save)
        out.append((mean.cpu(), dq.cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    qr = qb[:, :D].float().cpu()
    br = fb.dequantize()[:, :D].cpu()
    loss_ref = float(bank_loss.infonce_stats(qr, br, labels, tau)[2].mean())
    dq_ref = bank_loss.infonce_grad_q(qr, br, labels, tau)
    assert abs(float(out[0][0]) - loss_ref) < 2e-3 * max(1.0, abs(loss_ref))
    assert rel_err(out[0][1][:, :D], dq_ref) < 1.5e-2


@pytest.mark.parametrize("noise", [3.0, 3.5, 5.0])        # smallest 1 - p_label of the batch ~ 1e-4, 1e-3, 4e-2
def test_bank_confident_rows(ops, noise):
    """Queries that (nearly) coincide with their target row, p_label -> 1: the gradient is the small difference
    sum_j p_j bank_j - bank_label.  Both formulations keep it to bf16 accuracy of G (the two-pass kernels round
    G = p - onehot; the fused pass keeps the label key out of its accumulator and the fold applies (1 - p_label) in fp32).
    Below 1 - p_label ~ 1e-4 the fp32 lse itself (ulp 2e-6 at lse ~ 20) bounds every fp32 formulation, torch's included:
    measured at 1 - p = 2e-7: fused 1.06, two-pass 2.3, torch fp32 autograd 0.45 relative row error."""
    from oracle import bank_loss
    B, M, D, tau = 32, 6000, 768, 0.02
    g = torch.Generator().manual_seed(5)
    bank = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1)
    labels = torch.randint(0, M, (B,), generator=g)
    text = bank[labels] + noise * torch.randn(B, D, generator=g) / D ** 0.5
    _, qb, _ = ops.combine_l2norm_fwd(None, None, dev(text))
    bank_b = ops.prepare_bank(dev(bank))
    qr, br = qb.cpu().float(), bank_b.cpu().float()
    dq_ref = bank_loss.infonce_grad_q(qr, br, labels, tau)
    lse_ref, _, _ = bank_loss.infonce_stats(qr, br, labels, tau)
    save = ops.bank_logits_buffer(B, M, "cuda")
    stats = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau, save=save)
    lse, _, _ = ops.bank_loss_finalize(stats, M)
    assert (lse.cpu().double() - lse_ref).abs().max() < 2e-4
    def row_err(d):                                  # per query: the confident rows are not hidden behind the others' scale
        d = d.double().cpu()
        return ((d - dq_ref).abs().amax(1) / dq_ref.abs().amax(1).clamp_min(1e-30)).max().item()
    assert row_err(ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B, saved=save)) < 3e-2
    assert row_err(ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B)) < 3e-2


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "e4m3"])
@pytest.mark.parametrize("B,M,D", [(32, 40000, 768), (16, 200000, 768), (100, 9001, 1024), (40, 12000, 256)])
def test_bank_stream_kernels_race_screen(ops, B, M, D, fp8):
    """The stream kernels order their LDS-DMA tile ring, the partial-sum exchange and the asm transposed reads by counted
    `vmcnt` / hand-counted `lgkmcnt` waits and RAW barriers (no compiler fence).  Screen for races: 24 launches of the fused
    pair and of the two-pass pair, half of them while a bandwidth-heavy copy runs on a second stream, must all be
    bit-identical to the first."""
    g = torch.Generator().manual_seed(B + M + D)
    bank = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=-1)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1)
    labels = dev(torch.randint(0, M, (B,), generator=g))
    _, qb, _ = ops.combine_l2norm_fwd(None, None, dev(q))
    bank_b = ops.prepare_bank(dev(bank), "fp8" if fp8 else "bf16")
    save = ops.bank_logits_buffer(B, M, "cuda")

    def run():
        st = ops.bank_stats_fwd(qb, bank_b, labels, 50.0, save=save)
        lse, _, _ = ops.bank_loss_finalize(st, M)
        dq = ops.bank_grad_q(qb, bank_b, labels, 50.0, lse, 1.0 / B, saved=save)
        st2 = ops.bank_stats_fwd(qb, bank_b, labels, 50.0)
        dq2 = ops.bank_grad_q(qb, bank_b, labels, 50.0, lse, 1.0 / B)
        return st, dq, st2, dq2

    first = [t.clone() for t in run()]
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    for it in range(24):
        if it % 2:
            with torch.cuda.stream(side):
                big.add_(1.0)
        for a, b in zip(run(), first):
            assert torch.equal(a, b), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,M,D,tau", [(32, 4099, 512, 0.02), (70, 3000, 768, 0.02), (127, 9001, 1024, 0.05), (3, 33, 256, 0.01)])
def test_bank_fused_e4m3_image_kernel(ops, B, M, D, tau):
    """spn_bank_config(4): the fused pass over an e4m3 bank on the kernel that dequantises each raw tile into a bf16 image
    for the dq GEMM (logits still fp8 x fp8) - the path of chunks longer than the 2 048-entry row-scale table."""
    from oracle import bank_loss
    from spn4cir_amd import _lib
    _lib.lib().spn_bank_config(4)
    try:
        _saved_pair_case(ops, bank_loss, B, M, D, tau, True, check_z=False)
    finally:
        _lib.lib().spn_bank_config(0)


def _saved_pair_case(ops, bank_loss, B, M, D, tau, fp8, check_z=True, split_q=True):
    text, refer, bank, ridx, labels = _bank_case(B, M, D, 3 * B + M)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    bank_b = ops.prepare_bank(dev(bank), "fp8" if fp8 else "bf16")
    qr = qb[:, :D].cpu().float()
    br = (bank_b.dequantize() if fp8 else bank_b)[:, :D].cpu().float()
    if fp8 and not split_q:
        br = br.bfloat16().float()               # kernels that multiply the bf16 image of the dequantised tile
    if fp8 and split_q:          # the logits of an e4m3 bank run on the fp8 MFMA: queries as two e4m3 terms (oracle: split_query_e4m3)
        qr = bank_loss.split_query_e4m3(qr)   # (so does the fused pass; its dq GEMM reads a dequantised bf16 image)
    lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(qr, br, labels, tau)
    save = ops.bank_logits_buffer(B, M, "cuda")
    assert save is not None
    stats = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau, save=save)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    tol = 1e-3 if fp8 else 2e-4                   # e4m3: the two-term query model reproduces the logits to fp32 rounding of
    assert (lse.cpu().double() - lse_ref).abs().max() < tol               # values up to 1 / tau
    assert (stats[:, 3].cpu().double() - lab_ref).abs().max() < tol
    assert abs(mean.item() - row_ref.mean().item()) < tol
    if check_z:
        ld = (M + 31) // 32 * 32
        z = save[:B * ld * 4].view(torch.float32).view(B, ld)[:, :M].cpu().double()
        assert (z - (qr.double() @ br.double().t()) / tau).abs().max() < (5e-3 if fp8 else 5e-4)   # fp32 products of logits up to 1 / tau
    dq = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B, saved=save)
    dq_ref = bank_loss.infonce_grad_q(qr, br, labels, tau)
    assert rel_err(dq[:, :D], dq_ref) < 1.5e-2     # G is rounded to bf16 before the second GEMM
    assert not dq[:, D:].any()
    dq_old = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B)          # recomputing kernel
    assert rel_err(dq, dq_old) < 1.5e-2
    if M >= 64:
        eps, cut = 0.1, (M // 3) | 1                 # an odd cut: shards that start / end inside a tile
        qd = qr.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy((qd @ br.double().t()) / tau, labels, label_smoothing=eps)
        ref.backward()
        halves = []
        for a, b in ((0, cut), (cut, M)):
            if fp8:
                sh = ops.Fp8Bank(bank_b.data[a:b].contiguous(), bank_b.scale[a:b].contiguous())
            else:
                sh = bank_b[a:b].contiguous()
            sv = ops.bank_logits_buffer(B, b - a, "cuda")
            halves.append((sh, sv, a, ops.bank_stats_fwd(qb, sh, dev(labels), 1.0 / tau, m_begin=a, save=sv)))
        lse2, row2, mean2 = ops.bank_loss_finalize(torch.stack([h[3] for h in halves]), M, label_smoothing=eps)
        assert abs(mean2.item() - ref.item()) < 3e-4
        d = sum(ops.bank_grad_q(qb, sh, dev(labels), 1.0 / tau, lse2, 1.0 / B, M_total=M, label_smoothing=eps, m_begin=a,
                                saved=sv) for sh, sv, a, _ in halves)
        assert rel_err(d[:, :D], qd.grad) < 1.5e-2
        # the same shards without label smoothing: every shard's backward call uses its own saved buffer and the GLOBAL lse
        lse0, _, _ = ops.bank_loss_finalize(torch.stack([h[3] for h in halves]), M)
        d0 = sum(ops.bank_grad_q(qb, sh, dev(labels), 1.0 / tau, lse0, 1.0 / B, M_total=M, m_begin=a, saved=sv)
                 for sh, sv, a, _ in halves)
        assert rel_err(d0[:, :D], dq_ref) < 1.5e-2


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "e4m3"])
@pytest.mark.parametrize("B,M,D,tau", [(32, 40000, 768, 0.02), (4, 500, 128, 0.01), (33, 1500, 640, 0.02), (1, 7, 128, 0.02),
                                        (127, 9001, 1024, 0.05), (160, 40000, 768, 0.02), (32, 100000, 768, 0.02),
                                        (128, 30000, 256, 0.03),
                                        # 192..256 queries: the GEMM-shaped pair behind one call (statistics pass + ONE tail launch for
                                        # fold / finalize / mean, G^T scaling, dq GEMM); ragged bank sizes, a width other than 768
                                        (256, 40000, 768, 0.02), (192, 5000, 768, 0.02), (200, 4001, 512, 0.02), (256, 3000, 1024, 0.05)])
def test_bank_step_single_call(ops, B, M, D, tau, fp8):
    """spn_bank_step (one pass over the bank + ONE tail launch) against the three-call path it replaces
    (spn_bank_stats_fwd_save + spn_bank_loss_finalize + spn_bank_grad_q_saved): lse and dq BIT-identical (a bank-sharded
    data-parallel step must keep reproducing the single-process one), row loss / mean to 2e-5, the oracle on the same rounded
    operands; the ticketed mean is bit-reproducible from call to call."""
    from oracle import bank_loss
    text, refer, bank, ridx, labels = _bank_case(B, M, D, 7 * B + M)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    bank_b = ops.prepare_bank(dev(bank), "fp8" if fp8 else "bf16")
    Dp = qb.shape[1]
    if not ops.bank_step_ok(B, M, Dp, bank_b):
        pytest.skip("shape not served by the single-pass kernels")
    lab = dev(labels)
    save = ops.bank_logits_buffer(B, M, "cuda")
    lse, row, mean, dq = ops.bank_step(qb, bank_b, lab, 1.0 / tau, 1.0 / B, save)
    save2 = ops.bank_logits_buffer(B, M, "cuda")
    stats = ops.bank_stats_fwd(qb, bank_b, lab, 1.0 / tau, save=save2)
    lse3, row3, mean3 = ops.bank_loss_finalize(stats, M)
    dq3 = ops.bank_grad_q(qb, bank_b, lab, 1.0 / tau, lse3, 1.0 / B, saved=save2)
    assert torch.equal(lse, lse3) and torch.equal(dq, dq3)           # the same arithmetic in the same order
    if B >= 192 and not fp8:                                         # the large path repeats the finalize kernel's tree: bit-identical
        assert torch.equal(row, row3) and torch.equal(mean, mean3)
    assert (row - row3).abs().max() < 2e-5
    assert abs(mean.item() - mean3.item()) < 2e-5 * max(1.0, abs(mean3.item()))
    if not fp8:      # oracle on the operands the kernels see (the e4m3 kernels' operand model: test_bank_fused_single_pass)
        qr, br = qb[:, :D].cpu().float(), bank_b[:, :D].cpu().float()
        lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(qr, br, labels, tau)
        assert (lse.cpu().double() - lse_ref).abs().max() < 2e-4
        assert abs(mean.item() - row_ref.mean().item()) < 2e-4
        dq_ref = bank_loss.infonce_grad_q(qr, br, labels, tau)
        assert rel_err(dq[:, :D], dq_ref) < 1.5e-2
    # run-to-run: the ticketed mean and every output are bit-identical
    lse_b, row_b, mean_b, dq_b = ops.bank_step(qb, bank_b, lab, 1.0 / tau, 1.0 / B, save)
    assert torch.equal(mean, mean_b) and torch.equal(row, row_b) and torch.equal(dq, dq_b)


@pytest.mark.parametrize("mode", [0, 2], ids=["gemm-pair", "fused"])
@pytest.mark.parametrize("B,M,D,tau", [(256, 40000, 768, 0.02), (128, 3000, 256, 0.03), (264, 5001, 512, 0.02), (192, 5000, 768, 0.02),
                                        (512, 8000, 768, 0.02), (256, 3000, 1024, 0.05)])
def test_bank_saved_pair_large_batch(ops, B, M, D, tau, mode):
    """B >= 128: the forward GEMM keeps p = exp(logit - tile max) and the backward pass is G^T + ONE weight-gradient-shaped
    GEMM (spn_bank_grad_q_saved): dq against the oracle and against the recomputing kernel, with label smoothing and
    two unequal shards (global labels, m_begin).  mode 2: the fused single pass at the same shapes."""
    from oracle import bank_loss
    from spn4cir_amd import _lib
    _lib.lib().spn_bank_config(mode)
    try:
        _large_pair_case(ops, bank_loss, B, M, D, tau)
    finally:
        _lib.lib().spn_bank_config(0)


def _large_pair_case(ops, bank_loss, B, M, D, tau):
    text, refer, bank, ridx, labels = _bank_case(B, M, D, 5 * B + M)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    bank_b = ops.prepare_bank(dev(bank))
    qr, br = qb[:, :D].cpu().float(), bank_b[:, :D].cpu().float()
    lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(qr, br, labels, tau)
    save = ops.bank_logits_buffer(B, M, "cuda")
    assert save is not None
    stats = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau, save=save)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    assert (lse.cpu().double() - lse_ref).abs().max() < 2e-4
    assert abs(mean.item() - row_ref.mean().item()) < 2e-4
    dq = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B, saved=save)
    assert rel_err(dq[:, :D], bank_loss.infonce_grad_q(qr, br, labels, tau)) < 2e-2      # p and G each rounded to bf16
    assert rel_err(dq, ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B)) < 2e-2
    eps, cut = 0.1, (M // 3) | 1
    qd = qr.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy((qd @ br.double().t()) / tau, labels, label_smoothing=eps)
    ref.backward()
    halves = []
    for a, b in ((0, cut), (cut, M)):
        sh = bank_b[a:b].contiguous()
        sv = ops.bank_logits_buffer(B, b - a, "cuda")
        halves.append((sh, sv, a, ops.bank_stats_fwd(qb, sh, dev(labels), 1.0 / tau, m_begin=a, save=sv)))
    lse2, row2, mean2 = ops.bank_loss_finalize(torch.stack([h[3] for h in halves]), M, label_smoothing=eps)
    assert abs(mean2.item() - ref.item()) < 3e-4
    d = sum(ops.bank_grad_q(qb, sh, dev(labels), 1.0 / tau, lse2, 1.0 / B, M_total=M, label_smoothing=eps, m_begin=a,
                            saved=sv) for sh, sv, a, _ in halves)
    assert rel_err(d[:, :D], qd.grad) < 2e-2


def test_bank_label_smoothing_and_shards(ops):
    from oracle import bank_loss
    B, M, D, tau, eps = 24, 2500, 256, 0.02, 0.1
    text, refer, bank, ridx, labels = _bank_case(B, M, D, 77)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    bank_b = ops.prepare_bank(dev(bank))
    qr, br = qb.cpu().float(), bank_b.cpu().float()
    ref = torch.nn.functional.cross_entropy((qr.double() @ br.double().t()) / tau, labels, label_smoothing=eps)
    # two unequal shards, as two ranks would hold them
    cut = 1000
    s0 = ops.bank_stats_fwd(qb, bank_b[:cut].contiguous(), dev(labels), 1.0 / tau, m_begin=0)
    s1 = ops.bank_stats_fwd(qb, bank_b[cut:].contiguous(), dev(labels), 1.0 / tau, m_begin=cut)
    lse, row, mean = ops.bank_loss_finalize(torch.stack([s0, s1]), M, label_smoothing=eps)
    assert abs(mean.item() - ref.item()) < 3e-4
    qd = qr.double().requires_grad_(True)
    torch.nn.functional.cross_entropy((qd @ br.double().t()) / tau, labels, label_smoothing=eps).backward()
    d0 = ops.bank_grad_q(qb, bank_b[:cut].contiguous(), dev(labels), 1.0 / tau, lse, 1.0 / B, M_total=M,
                         label_smoothing=eps, m_begin=0)
    d1 = ops.bank_grad_q(qb, bank_b[cut:].contiguous(), dev(labels), 1.0 / tau, lse, 1.0 / B, M_total=M,
                         label_smoothing=eps, m_begin=cut)
    assert rel_err(d0 + d1, qd.grad) < 1.5e-2


def test_bank_shards_large_batch_gemm_path(ops):
    """The per-rank shape of the data-parallel step (BASELINE configs 3-5): ALL gathered queries (B >= 128, so the forward
    runs on the GEMM path) against ONE contiguous shard of the bank, labels in global row numbers; the shards' statistics
    and query gradients combine to the single-bank result."""
    B, M, D, tau = 512, 8000, 768, 0.02
    text, refer, bank, ridx, labels = _bank_case(B, M, D, 4242)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    bank_b = ops.prepare_bank(dev(bank))
    qr, br = qb[:, :D].cpu().float(), bank_b[:, :D].cpu().float()
    qd = qr.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy((qd @ br.double().t()) / tau, labels)
    ref.backward()
    cuts = [0, 3000, 3001, 8000]                       # a one-row shard in the middle
    parts = [ops.bank_stats_fwd(qb, bank_b[a:b].contiguous(), dev(labels), 1.0 / tau, m_begin=a)
             for a, b in zip(cuts[:-1], cuts[1:])]
    lse, row, mean = ops.bank_loss_finalize(torch.stack(parts), M)
    assert abs(mean.item() - ref.item()) < 3e-4
    whole = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau)
    lse1, row1, mean1 = ops.bank_loss_finalize(whole, M)
    assert (lse - lse1).abs().max().item() < 2e-4
    dq = sum(ops.bank_grad_q(qb, bank_b[a:b].contiguous(), dev(labels), 1.0 / tau, lse, 1.0 / B, M_total=M, m_begin=a)
             for a, b in zip(cuts[:-1], cuts[1:]))
    assert rel_err(dq[:, :D], qd.grad) < 1.5e-2


def test_bank_matches_golden_loss_cases(ops, golden_dir):
    import os
    from cases import LOSS_CASES, loss_case_inputs
    z = np.load(os.path.join(golden_dir, "loss_cases.npz"))
    for ci in range(len(LOSS_CASES)):
        text, rb, bank, ridx, labels, tau = loss_case_inputs(ci)
        q, qb, inv = ops.combine_l2norm_fwd(dev(rb), dev(ridx), dev(text))
        bank_b = ops.prepare_bank(dev(bank))
        stats = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau)
        lse, row, mean = ops.bank_loss_finalize(stats, bank.shape[0])
        ref = float(z[f"c{ci}_loss"])
        assert abs(mean.item() - ref) < 1e-2 * max(1.0, abs(ref))
        dq = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / text.shape[0])
        dtext = ops.combine_l2norm_bwd(q, inv, dq[:, :text.shape[1]].contiguous())
        assert rel_err(dtext, torch.from_numpy(z[f"c{ci}_dtext"])) < 3e-2


@pytest.mark.parametrize("B,M,D,temp", [(4, 7, 256, 0.07), (32, 999, 256, 0.07), (128, 6000, 256, 0.05),
                                         (45, 300, 128, 0.03), (1, 1, 256, 0.07)])
def test_bank_tokmax(ops, B, M, D, temp):
    """SURVEY 8f-4 (BLIP-2 head): logit = max over the 32 token rows of a target; first-index ties as torch.max."""
    from oracle import bank_loss
    g = torch.Generator().manual_seed(B * 7 + M)
    bank = torch.nn.functional.normalize(torch.randn(M, 32, D, generator=g), dim=-1)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1)
    labels = torch.randint(0, M, (B,), generator=g)
    labels[0], labels[-1] = 0, M - 1
    bank_b = bank.to(torch.bfloat16).cuda().contiguous()
    qb = q.to(torch.bfloat16).cuda().contiguous()
    qr, br = qb.cpu().double(), bank_b.cpu().double()
    # reference on the same bf16-rounded operands (fp64): the oracle's per-sample loop
    qd = qr.clone().requires_grad_(True)
    ref = bank_loss.tokmax_infonce(qd, br, labels, temp)
    ref.backward()
    stats = ops.bank_stats_fwd_tokmax(qb, bank_b, dev(labels), 1.0 / temp)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    assert abs(mean.item() - ref.item()) < 2e-4 * max(1.0, abs(ref.item()))
    sim = torch.einsum("bd,mkd->bmk", qr, br).amax(-1) / temp
    assert (lse.cpu().double() - torch.logsumexp(sim, 1)).abs().max() < 2e-4
    assert (stats[:, 3].cpu().double() - sim[torch.arange(B), labels]).abs().max() < 2e-4
    dq = ops.bank_grad_q_tokmax(qb, bank_b, dev(labels), 1.0 / temp, lse, 1.0 / B)
    assert rel_err(dq, qd.grad) < 1.5e-2
    # shards of targets, as two ranks would hold them
    if M >= 4:
        cut = M // 3
        s0 = ops.bank_stats_fwd_tokmax(qb, bank_b[:cut].contiguous(), dev(labels), 1.0 / temp, t_begin=0)
        s1 = ops.bank_stats_fwd_tokmax(qb, bank_b[cut:].contiguous(), dev(labels), 1.0 / temp, t_begin=cut)
        lse2, _, mean2 = ops.bank_loss_finalize(torch.stack([s0, s1]), M)
        assert abs(mean2.item() - mean.item()) < 1e-5 and (lse2 - lse).abs().max() < 1e-5
        d0 = ops.bank_grad_q_tokmax(qb, bank_b[:cut].contiguous(), dev(labels), 1.0 / temp, lse, 1.0 / B,
                                    targets_total=M, t_begin=0)
        d1 = ops.bank_grad_q_tokmax(qb, bank_b[cut:].contiguous(), dev(labels), 1.0 / temp, lse, 1.0 / B,
                                    targets_total=M, t_begin=cut)
        assert rel_err(d0 + d1, dq) < 1e-5


def test_bank_tokmax_ties_and_head(ops):
    """Exact ties (duplicated token rows) send the gradient to the FIRST arg-max row; loss_qtc wires the autograd
    graph incl. the learnable temperature (blip2_qformer_cir_align_prompt.py:253-268)."""
    from oracle import bank_loss
    from spn4cir_amd import blip2_head
    g = torch.Generator().manual_seed(5)
    B, M, D, temp = 8, 40, 256, 0.07
    bank = torch.nn.functional.normalize(torch.randn(M, 32, D, generator=g), dim=-1).to(torch.bfloat16).float()
    bank[:, 17] = bank[:, 3]                     # every target: rows 3 and 17 identical
    bank[:, 29] = bank[:, 3]
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1).to(torch.bfloat16).float()
    labels = torch.randint(0, M, (B,), generator=g)
    qd = q.double().requires_grad_(True)
    td = torch.tensor(temp, dtype=torch.float64, requires_grad=True)
    ref = bank_loss.tokmax_infonce(qd, bank.double(), labels, td)
    ref.backward()
    qg = q.cuda().requires_grad_(True)
    tg = torch.tensor(temp, device="cuda", requires_grad=True)
    out = blip2_head.loss_qtc(qg, blip2_head.prepare_token_bank(bank), labels, tg)
    assert set(out) == {"loss_qtc"} and out["loss_qtc"].dim() == 0
    out["loss_qtc"].backward()
    assert abs(out["loss_qtc"].item() - ref.item()) < 2e-4
    assert rel_err(qg.grad, qd.grad) < 1.5e-2
    assert abs(tg.grad.item() - td.grad.item()) < 2e-2 * abs(td.grad.item()) + 1e-4
    # banks beyond one call's 4 GiB window go through in target shards: force 3 shards and expect the same numbers
    old = blip2_head._MAX_SHARD_BYTES
    try:
        blip2_head._MAX_SHARD_BYTES = 15 * 32 * 256 * 2
        q2 = q.cuda().requires_grad_(True)
        out2 = blip2_head.loss_qtc(q2, blip2_head.prepare_token_bank(bank), labels, temp)
        out2["loss_qtc"].backward()
    finally:
        blip2_head._MAX_SHARD_BYTES = old
    assert abs(out2["loss_qtc"].item() - out["loss_qtc"].item()) < 1e-5
    assert rel_err(q2.grad, qg.grad) < 1e-5


@pytest.mark.parametrize("tag", ["small", "m1000"])
def test_blip2_stage2_loss_matches_reference(ops, golden_dir, tag):
    """blip2_head.loss_qtc behind the reference's own query producer tail (text_proj_q + F.normalize of position 32) against
    loss and gradients captured from the reference's forward_stage2 (blip2_qformer_cir_align_prompt.py:247-268,
    tests/golden/make_golden_blip2.py): loss, d hidden[:, 32], d temp, d text_proj_q - incl. tied token rows."""
    from cases import blip2_target_feats
    from spn4cir_amd import blip2_head
    z = np.load(os.path.join(golden_dir, "blip2_stage2.npz"))
    get = lambda k: torch.from_numpy(np.asarray(z[f"{tag}.{k}"]))
    hidden = get("hidden").cuda().requires_grad_(True)
    proj = torch.nn.Linear(hidden.shape[-1], 256).cuda()
    with torch.no_grad():
        proj.weight.copy_(get("proj_w")); proj.bias.copy_(get("proj_b"))
    temp = torch.nn.Parameter(torch.tensor(float(z[f"{tag}.temp"]), device="cuda"))
    bank = blip2_head.prepare_token_bank(blip2_target_feats(tag))
    feats = torch.nn.functional.normalize(proj(hidden[:, 32, :]), dim=-1)
    assert (feats.detach().cpu() - get("fusion_feats")).abs().max() < 1e-5
    out = blip2_head.loss_qtc(feats, bank, get("target_indexs"), temp)
    out["loss_qtc"].backward()
    ref_loss = float(z[f"{tag}.loss_qtc"])
    assert abs(out["loss_qtc"].item() - ref_loss) < 1e-2 * max(1.0, abs(ref_loss))       # bf16 bank / queries
    assert rel_err(hidden.grad[:, 32, :], get("d_hidden32").cuda()) < 3e-2
    assert rel_err(proj.weight.grad, get("d_proj_w").cuda()) < 3e-2
    assert rel_err(proj.bias.grad, get("d_proj_b").cuda()) < 3e-2
    dt = float(z[f"{tag}.d_temp"])
    assert abs(temp.grad.item() - dt) < 3e-2 * abs(dt) + 1e-4


@pytest.mark.parametrize("B,M,D,tau", [(32, 4099, 512, 0.02), (16, 100000, 768, 0.02), (256, 40000, 768, 0.02),
                                        (33, 1500, 640, 0.03)])
def test_bank_fp8(ops, B, M, D, tau):
    """BASELINE config 5: the bank stored as e4m3 + per-row scale.  (1) the quantised bytes and scales are
    bit-identical to the oracle's; (2) against the oracle on the SAME dequantised operands only the
    accumulation order differs (2e-4 on lse / loss, 1.5e-2 relative on dq as for the bf16 bank); (3) against
    the bf16-bank path the survey's gate |delta loss| <= 1e-2."""
    from oracle import bank_loss
    text, refer, bank, ridx, labels = _bank_case(B, M, D, B + M + 5)
    q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
    fb = ops.prepare_bank(dev(bank), dtype="fp8")
    data_ref, scale_ref = bank_loss.quantize_e4m3(bank)
    assert torch.equal(fb.data[:, :D].cpu(), data_ref) and not fb.data[:, D:].any()
    assert torch.equal(fb.scale.cpu(), scale_ref)
    qr = qb[:, :D].cpu().float()
    br = bf(bank_loss.dequantize_e4m3(data_ref, scale_ref)).float()       # the dequantising kernels round the tile to bf16
    if B < 128:     # streaming forward on the fp8 MFMA: exact e4m3 bank x scale, queries as two e4m3 terms
        lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(bank_loss.split_query_e4m3(qr),
                                                            bank_loss.dequantize_e4m3(data_ref, scale_ref), labels, tau)
    else:           # large batches: the shard is expanded to bf16 once and the bf16 GEMM path runs on it
        lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(qr, br, labels, tau)
    stats = ops.bank_stats_fwd(qb, fb, dev(labels), 1.0 / tau)
    lse, row, mean = ops.bank_loss_finalize(stats, M)
    assert (lse.cpu().double() - lse_ref).abs().max() < 2e-4
    assert (stats[:, 3].cpu().double() - lab_ref).abs().max() < 2e-4
    assert abs(mean.item() - row_ref.mean().item()) < 2e-4
    dq = ops.bank_grad_q(qb, fb, dev(labels), 1.0 / tau, lse, 1.0 / B)
    assert rel_err(dq[:, :D], bank_loss.infonce_grad_q(qr, br, labels, tau)) < 1.5e-2
    # versus the bf16 bank
    bank_b = ops.prepare_bank(dev(bank))
    s16 = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau)
    lse16, row16, mean16 = ops.bank_loss_finalize(s16, M)
    assert abs(mean.item() - mean16.item()) < 1e-2 * max(1.0, abs(mean16.item()))
    dq16 = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse16, 1.0 / B)
    assert rel_err(dq[:, :D], dq16[:, :D]) < 0.15       # e4m3 keeps 3 mantissa bits of every bank element


# ----------------------------------------------------------------------------------- AdamW
def test_adamw_matches_golden(ops, golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "adamw.npz"))
    p = dev(torch.from_numpy(z["p0"]).clone())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    ops.adamw_step(p, dev(torch.from_numpy(z["g1"])), m, v, 1, float(z["lr"]))
    assert torch.allclose(p.cpu(), torch.from_numpy(z["p1"]), atol=2e-7, rtol=1e-6)
    # second step through the GradScaler path: scaled grads + inv_scale, found_inf = 0
    found = torch.zeros(1, device="cuda")
    g2 = dev(torch.from_numpy(z["g2"])) * 1024.0
    ops.grad_check_finite(g2, found)
    ops.adamw_step(p, g2, m, v, 2, float(z["lr"]), inv_scale=1.0 / 1024.0, found_inf=found)
    assert torch.allclose(p.cpu(), torch.from_numpy(z["p2"]), atol=2e-7, rtol=1e-6)
    # an inf gradient skips the step
    before = p.clone()
    g2[5] = float("inf")
    ops.grad_check_finite(g2, found)
    assert found.item() == 1.0
    ops.adamw_step(p, g2, m, v, 3, float(z["lr"]), found_inf=found)
    assert torch.equal(p, before)


# -------------------------------------------------------------------------------- Recall@K
def test_topk_identical_sets(ops, golden_dir):
    import json, os
    from oracle import recall
    z = np.load(os.path.join(golden_dir, "recall.npz"))
    pred, gallery = torch.from_numpy(z["pred"]), torch.from_numpy(z["gallery"])
    gn, _, _ = ops.combine_l2norm_fwd(None, None, dev(gallery))
    scores = ops.cosine_scores_f64(dev(pred), gn)
    idx, val = ops.topk_from_scores(scores, 50)
    order, sc = recall.ranked_indices(pred.numpy(), gallery.numpy())
    assert np.abs(scores.cpu().numpy() - sc).max() < 1e-6
    got = idx.cpu().numpy()
    assert all(set(got[i]) == set(z["top50"][i]) for i in range(got.shape[0]))       # vs the reference
    assert (got == order[:, :50]).mean() > 0.999                                      # vs the oracle, in order
    # exclusion of the reference image (validate.py:39)
    ex = torch.from_numpy(z["ref_idx"].astype(np.int32))
    idx2, _ = ops.topk_from_scores(scores, 10, exclude=dev(ex))
    got2 = idx2.cpu().numpy()
    for i in range(got2.shape[0]):
        row = [j for j in order[i] if j != int(ex[i])][:10]
        assert list(got2[i]) == row


@pytest.mark.parametrize("Ng,K", [(6000, 50), (40000, 50), (300, 256), (37, 50), (5000, 1), (1000, 10)])
def test_topk_ties_and_edges(ops, Ng, K):
    """Top-K selection (radix select on (score, ~index)): heavy ties (scores drawn from 7 distinct values, so the K-th score
    is always tied), ascending-index tie rule, an excluded column per row, fewer candidates than K (-1 fill), K = 1."""
    g = torch.Generator().manual_seed(Ng + K)
    Nq = 9
    vals = torch.tensor([-0.5, -0.25, 0.0, 0.125, 0.3, 0.3000000001, 0.9], dtype=torch.float64)
    scores = vals[torch.randint(0, 7, (Nq, Ng), generator=g)]
    scores[1] = 0.25                                           # a whole row of equal scores
    scores[2] = torch.randn(Ng, generator=g, dtype=torch.float64)     # and one without ties
    ex = torch.randint(0, Ng, (Nq,), generator=g).to(torch.int32)
    for exclude in (None, ex):
        idx, val = ops.topk_from_scores(dev(scores), K, exclude=None if exclude is None else dev(exclude))
        idx, val = idx.cpu(), val.cpu()
        for i in range(Nq):
            order = sorted((j for j in range(Ng) if exclude is None or j != int(exclude[i])),
                           key=lambda j: (-scores[i, j].item(), j))[:K]
            want = order + [-1] * (K - len(order))
            assert idx[i].tolist() == want, (i, Ng, K)
            assert torch.equal(val[i, :len(order)], scores[i, order])


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 128), (4096, 768, 768), (19712, 768, 3072), (1000, 260, 192)])
def test_gemm_nt_phased_schedule_race_screen(ops, M, N, K):
    """The staggered 8-slot schedule orders LDS-DMA writes and ds_reads only by counted vmcnt + barriers.
    Screen for rare races: 40 launches, half of them while a bandwidth-heavy copy runs on a second stream
    (perturbs DMA latency), must all be bit-identical to the first and match the fp64 product of the same
    bf16 operands (fp32 accumulation: 2e-3 relative to the largest output)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g)).cuda()
    b = bf(torch.randn(N, K, generator=g)).cuda()
    first = ops.gemm_nt(a, b, out_dtype=torch.float32)
    ref = (a.double() @ b.double().t())
    assert ((first.double() - ref).abs().max() / ref.abs().max()).item() < 2e-3
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    for it in range(40):
        if it % 2:
            with torch.cuda.stream(side):
                big.add_(1.0)
        out = ops.gemm_nt(a, b, out_dtype=torch.float32)
        assert torch.equal(out, first), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("Kr,N1,N2", [(64, 768, 2304), (200, 768, 768), (4096, 768, 3072), (19712, 3072, 768), (1000, 1032, 2056)])
def test_gemm_tn_phased_schedule_race_screen(ops, Kr, N1, N2):
    """Same screen for the weight-gradient kernel (split-K partials + reduction are deterministic)."""
    g = torch.Generator().manual_seed(Kr + N1 + N2)
    a = bf(torch.randn(Kr, N1, generator=g)).cuda()
    b = bf(torch.randn(Kr, N2, generator=g)).cuda()
    first = ops.gemm_tn(a, b)
    ref = (a.double().t() @ b.double())
    assert ((first.double() - ref).abs().max() / ref.abs().max()).item() < 2e-3
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    for it in range(30):
        if it % 2:
            with torch.cuda.stream(side):
                big.add_(1.0)
        out = ops.gemm_tn(a, b)
        assert torch.equal(out, first), it
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------- seeded shape sweeps
def test_shape_sweep_gemm(ops):
    """40 seeded random shapes through both GEMM kernels (ragged M / N / Kr, every K multiple of 64, duplicate rows):
    against the fp64 product of the same bf16 operands, 2e-3 relative to the largest output."""
    rng = np.random.default_rng(123)
    for it in range(40):
        M, N = int(rng.integers(1, 1400)), int(rng.integers(1, 200)) * 8
        K = int(rng.integers(1, 20)) * 64
        g = torch.Generator().manual_seed(it)
        a, b = bf(torch.randn(M, K, generator=g)).cuda(), bf(torch.randn(N, K, generator=g)).cuda()
        ref = a.double() @ b.double().t()
        out = ops.gemm_nt(a, b, out_dtype=torch.float32)
        assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 2e-3, ("nt", M, N, K)
        Kr, N1 = int(rng.integers(1, 3000)), int(rng.integers(1, 150)) * 8
        x, y = bf(torch.randn(Kr, N1, generator=g)).cuda(), bf(torch.randn(Kr, N, generator=g)).cuda()
        ref = x.double().t() @ y.double()
        out = ops.gemm_tn(x, y)
        assert ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item() < 2e-3, ("tn", Kr, N1, N)


def test_shape_sweep_bank(ops):
    """25 seeded (B, M, D, tau) draws incl. M not a multiple of the 32-row tile, M < 32, labels on the first / last
    row and duplicated labels: loss statistics and dq against the oracle on the same bf16 operands."""
    from oracle import bank_loss
    rng = np.random.default_rng(321)
    dims = [64, 128, 256, 512, 640, 768, 1024]
    for it in range(25):
        B, M = int(rng.integers(1, 300)), int(rng.integers(1, 9000))
        D, tau = int(dims[rng.integers(0, len(dims))]), float(rng.choice([0.01, 0.02, 0.03, 0.07]))
        text, refer, bank, ridx, labels = _bank_case(B, M, D, 1000 + it)
        if B > 2:
            labels[1] = labels[2]                      # duplicate targets inside a batch
        q, qb, inv = ops.combine_l2norm_fwd(dev(refer), dev(ridx), dev(text))
        bank_b = ops.prepare_bank(dev(bank))
        qr, br = qb[:, :D].cpu().float(), bank_b[:, :D].cpu().float()
        lse_ref, lab_ref, row_ref = bank_loss.infonce_stats(qr, br, labels, tau)
        stats = ops.bank_stats_fwd(qb, bank_b, dev(labels), 1.0 / tau)
        lse, row, mean = ops.bank_loss_finalize(stats, M)
        assert (lse.cpu().double() - lse_ref).abs().max() < 3e-4, (B, M, D, tau)
        assert abs(mean.item() - row_ref.mean().item()) < 3e-4, (B, M, D, tau)
        dq = ops.bank_grad_q(qb, bank_b, dev(labels), 1.0 / tau, lse, 1.0 / B)
        assert rel_err(dq[:, :D], bank_loss.infonce_grad_q(qr, br, labels, tau)) < 2e-2, (B, M, D, tau)


def test_output_canaries(ops):
    """Out-of-bounds screen: outputs are carved out of larger buffers filled with a sentinel;
    ragged shapes (rows / columns that do not fill a tile) must leave every byte outside the output untouched."""
    import ctypes as C
    from spn4cir_amd._lib import check, lib
    from spn4cir_amd.ops import _p, _stream, workspace
    SENT = -12345.0
    for (M, N, K) in [(1, 8, 64), (257, 264, 128), (1000, 260, 192), (300, 776, 64)]:
        g = torch.Generator().manual_seed(M)
        a, b = bf(torch.randn(M, K, generator=g)).cuda(), bf(torch.randn(N, K, generator=g)).cuda()
        pad = 4096
        buf = torch.full((pad + M * N + pad,), SENT, dtype=torch.float32, device="cuda")
        out = buf[pad:pad + M * N]
        check(lib().spn_gemm_nt(_p(a), _p(b), M, N, K, K, K, None, 0, None, _p(out), None, N, _stream()), "gemm_nt")
        torch.cuda.synchronize()
        assert (buf[:pad] == SENT).all() and (buf[pad + M * N:] == SENT).all(), ("nt f32", M, N, K)
        ref = a.double() @ b.double().t()
        assert ((out.view(M, N).double() - ref).abs().max() / ref.abs().max()).item() < 2e-3
        bufb = torch.full((pad + M * N + pad,), SENT, dtype=torch.bfloat16, device="cuda")
        outb = bufb[pad:pad + M * N]
        check(lib().spn_gemm_nt(_p(a), _p(b), M, N, K, K, K, None, 0, _p(outb), None, None, N, _stream()), "gemm_nt")
        torch.cuda.synchronize()
        sb = torch.tensor(SENT, dtype=torch.bfloat16).item()
        assert (bufb[:pad].float() == sb).all() and (bufb[pad + M * N:].float() == sb).all(), ("nt bf16", M, N, K)
    for (Kr, N1, N2) in [(5, 8, 8), (333, 264, 776), (2000, 776, 2312)]:
        g = torch.Generator().manual_seed(Kr)
        x, y = bf(torch.randn(Kr, N1, generator=g)).cuda(), bf(torch.randn(Kr, N2, generator=g)).cuda()
        pad = 4096
        buf = torch.full((pad + N1 * N2 + pad,), SENT, dtype=torch.float32, device="cuda")
        out = buf[pad:pad + N1 * N2].view(N1, N2)
        ops.gemm_tn(x, y, out=out)
        torch.cuda.synchronize()
        assert (buf[:pad] == SENT).all() and (buf[pad + N1 * N2:] == SENT).all(), ("tn", Kr, N1, N2)
        ref = x.double().t() @ y.double()
        assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 2e-3


def test_c_abi_error_codes(ops):
    """Boundary contract (include/spn4cir_hip.h): bad arguments come back as negative SPN_ERR_* codes - nothing throws
    inside the library, nothing is launched - and the Python side turns them into RuntimeError with the library's text."""
    from spn4cir_amd._lib import lib
    L = lib()
    a = torch.zeros(128, 96, dtype=torch.bfloat16, device="cuda")       # K = 96 is not a multiple of the 64-deep k step
    b = torch.zeros(128, 96, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.gemm_nt(a, b)
    out = torch.zeros(128, 128, dtype=torch.float32, device="cuda")
    p = lambda t: t.data_ptr()
    # gemm_tn: workspace too small -> SPN_ERR_WORKSPACE (-3); null output -> an error code, not a crash
    a2 = torch.zeros(256, 128, dtype=torch.bfloat16, device="cuda")
    rc = L.spn_gemm_tn(p(a2), p(a2), 256, 128, 128, 128, 128, p(out), 128, 1.0, 0, None, p(out), 16, None)
    assert rc == -3
    # bank: D outside the supported widths -> SPN_ERR_SHAPE; B = 0 -> SPN_ERR_ARG
    q = torch.zeros(4, 96, dtype=torch.bfloat16, device="cuda")
    bank = torch.zeros(8, 96, dtype=torch.bfloat16, device="cuda")
    lab = torch.zeros(4, dtype=torch.int64, device="cuda")
    st = torch.zeros(4, 4, device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    assert L.spn_bank_stats_fwd(p(q), 96, p(bank), p(lab), 4, 8, 96, 0, 50.0, p(st), p(ws), ws.numel(), None) == -2
    q2 = torch.zeros(4, 128, dtype=torch.bfloat16, device="cuda")
    bank2 = torch.zeros(8, 128, dtype=torch.bfloat16, device="cuda")
    assert L.spn_bank_stats_fwd(p(q2), 128, p(bank2), p(lab), 0, 8, 128, 0, 50.0, p(st), p(ws), ws.numel(), None) == -1
    assert L.spn_bank_stats_fwd(p(q2), 128, p(bank2), p(lab), 4, 8, 128, 0, 50.0, None, p(ws), ws.numel(), None) == -1
    # 160-row-tile statistics pass (128..256 queries): a workspace sized for the 256-row tiling only (ceil(M/256) blocks of
    # B*16 bytes) is too small for its ceil(M/160) partial blocks -> SPN_ERR_WORKSPACE before anything is launched
    Bq, Mq, Dq = 256, 4000, 256
    q3 = torch.zeros(Bq, Dq, dtype=torch.bfloat16, device="cuda")
    bank3 = torch.zeros(Mq, Dq, dtype=torch.bfloat16, device="cuda")
    lab3 = torch.zeros(Bq, dtype=torch.int64, device="cuda")
    st3 = torch.zeros(Bq, 4, device="cuda")
    small = ((Mq + 255) // 256) * Bq * 16
    if os.environ.get("SPN_BANK_S160", "1") != "0" and os.environ.get("SPN_BANK_GEMM", "1") != "0":
        assert L.spn_bank_stats_fwd(p(q3), Dq, p(bank3), p(lab3), Bq, Mq, Dq, 0, 50.0, p(st3), p(ws), small, None) == -3
    assert L.spn_bank_stats_fwd(p(q3), Dq, p(bank3), p(lab3), Bq, Mq, Dq, 0, 50.0, p(st3), p(ws),
                                L.spn_bank_workspace_bytes(Bq, Mq, Dq), None) == 0
    # token-max bank: rows must come in groups of 32 per target, widths as the plain bank
    assert L.spn_bank_stats_fwd_tokmax(p(q2), 128, p(bank2), p(lab), 4, 0, 128, 0, 14.0, p(st), p(ws), ws.numel(), None) == -1
    # TG-CIR head: more than 640 positions / 8 local tokens only
    z = torch.zeros(2, 4, 64, device="cuda")
    assert L.spn_tg_tokenlearn_fwd(p(z), p(z), p(z), p(z), p(z), 2, 700, 64, 8, 4, None) == -2
    assert L.spn_tg_tokenlearn_fwd(p(z), p(z), p(z), p(z), p(z), 2, 4, 64, 6, 4, None) == -2
    for code, text in ((-1, "invalid argument"), (-2, "unsupported shape"), (-3, "workspace")):
        assert text.split()[0] in L.spn_error_string(code).decode()
    # the library is still healthy afterwards
    assert rel_err(ops.gemm_nt(dev(bf(torch.eye(128))), dev(bf(torch.eye(128))), out_dtype=torch.float32), torch.eye(128)) < 1e-6


def test_gemm_nt3_hand_scheduled_kernel():
    """The experimental 4-wave NT GEMM with the generated inline-asm main loop (SPN_GEMM_CFG=7, tools/gen_gemm3.py):
    every epilogue mode against a torch fp32 product, incl. shapes that fall back to gemm_nt2 (partial tiles) and
    odd / even k-tile counts (the peeled loop tails).  Own process: the configuration is read once per process."""
    import os, re, subprocess, sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spn4cir_amd import _lib
    if not _lib.config_dump()["experiments_build"]:
        pytest.skip("gemm_nt3 is compiled only into the experiments build (tests/test_experiments_gpu.py runs this there)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPN_GEMM_CFG="7")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm3_check.py"), "child", "check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "check" in l]
    assert len(lines) >= 6 * 2 * 7, out.stdout[-2000:]
    bad = [l for l in lines if not re.search(r"bad elements 0$", l)]
    assert not bad, "\n".join(bad[:10])


@pytest.mark.parametrize("Kr,shapes", [
    (640, [(256, 512), (520, 264)]),                                   # fewer tiles than CUs: everything is "tail", split
    (448, [(768, 3072), (3072, 768), (2304, 768), (768, 768)] * 3),    # 324 tiles: a full round + a split tail of 68
    (384, [(2048, 2048)] * 4 + [(256, 256)] * 5),                      # 261 tiles: tail of 5 tiles, split 6 ways
    (200, [(264, 72), (8, 520)]),                                      # ragged tiles, Kr not a multiple of 64
])
def test_gemm_tn_grouped(ops, Kr, shapes):
    """Grouped weight gradients without split-K (spn_gemm_tn_grouped): every problem against a torch fp32 product,
    incl. the column sums; full rounds are written directly, the last partial round goes through slabs."""
    g = torch.Generator(device="cuda").manual_seed(Kr)
    pairs = [(torch.randn(Kr, n1, device="cuda", generator=g).bfloat16(), torch.randn(Kr, n2, device="cuda", generator=g).bfloat16())
             for n1, n2 in shapes]
    for rep in range(2):
        outs = ops.gemm_tn_grouped(pairs)
        for (a, b), (c, cs) in zip(pairs, outs):
            ref = a.float().t() @ b.float()
            assert (c - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-3, (tuple(a.shape), tuple(b.shape))
            rs = a.float().sum(0)
            assert (cs - rs).abs().max() <= 2e-3 * rs.abs().max() + 1e-2
    # transpose-detecting, asymmetric check of one problem against the per-problem kernel
    single = ops.gemm_tn(pairs[0][0], pairs[0][1])
    assert torch.allclose(outs[0][0], single, rtol=1e-4, atol=1e-3 * single.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("S,E,packed,L,H", [(70, 128, False, 9, 2), (197, 256, True, 9, 2), (577, 128, False, 9, 2), (577, 256, True, 9, 2),
                                            # several row tiles of the 8-wave kernels (96 rows; apply: 192 rows x 384 / 256 columns),
                                            # ragged last tiles, both column-tile widths of the apply kernel
                                            (577, 384, True, 70, 4), (577, 768, False, 25, 4), (130, 512, True, 40, 6)])
def test_xattn_absorbed_matches_torch(S, E, packed, L, H):
    """spn_xattn_fwd / spn_xattn_bwd (cross-attention over frozen tokens with the K/V projections absorbed into the query and
    output side, csrc/xattn.hip) against the textbook form in fp32 torch: K = X Wk^T + bk, V = X Wv^T + bv, softmax(q K^T / 8) V
    per head (blip4cir/med.py:196-234), autograd for dq, dWkv, dbkv.  Dense and packed rows, both column-tile instantiations."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spn4cir_amd import ops
    g = torch.Generator().manual_seed(S + E)
    B = 5
    W = H * 64
    lens = torch.tensor([L, 1, (4 * L) // 9, L, (2 * L) // 3]) if packed else torch.full((B,), L)
    rows = int(lens.sum())
    bf = torch.bfloat16
    q = (torch.randn(rows, W, generator=g)).to(bf)
    wkv = (torch.randn(2 * W, E, generator=g) * 0.15).to(bf)
    bkv = torch.randn(2 * W, generator=g) * 0.3
    x = torch.randn(B, S, E, generator=g).to(bf)
    dctx = torch.randn(rows, W, generator=g).to(bf)
    cu = None
    if packed:
        cu = torch.zeros(B + 1, dtype=torch.int32)
        cu[1:] = lens.cumsum(0)
    ctx, saved = ops.xattn_fwd(q.cuda(), wkv.cuda(), bkv.cuda(), x.cuda(), H, cu=None if cu is None else cu.cuda(), L=L)
    dq, dwkv, dbkv = ops.xattn_bwd(saved, dctx.cuda())
    # reference
    qf = q.float().requires_grad_(True)
    wf = wkv.float().requires_grad_(True)
    bfv = bkv.clone().requires_grad_(True)
    outs = []
    r0 = 0
    for b in range(B):
        n = int(lens[b])
        qb = qf[r0:r0 + n].view(n, H, 64).transpose(0, 1)                       # [H, n, 64]
        kv = x[b].float() @ wf.t() + bfv                                        # [S, 2W]
        k = kv[:, :W].view(S, H, 64).transpose(0, 1)
        v = kv[:, W:].view(S, H, 64).transpose(0, 1)
        p = torch.softmax(qb @ k.transpose(1, 2) * 0.125, dim=-1)
        outs.append((p @ v).transpose(0, 1).reshape(n, W))
        r0 += n
    ref = torch.cat(outs)
    ref.backward(dctx.float())

    def rel(a, b):
        return ((a.float().cpu() - b).norm() / b.norm()).item()
    assert rel(ctx, ref.detach()) < 1.5e-2
    assert rel(dq, qf.grad) < 2.5e-2
    assert rel(dwkv, wf.grad) < 2.5e-2
    assert rel(dbkv[W:], bfv.grad[W:]) < 1e-2
    assert dbkv[:W].abs().max().item() == 0.0                                    # softmax shift invariance: exactly zero
    assert bfv.grad[:W].abs().max().item() < 1e-4 * bfv.grad[W:].abs().max().item()
