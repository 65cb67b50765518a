"""Oracle: combiner + L2-normalise + scaled-negative bank InfoNCE (CPU fp32).

Restates clip4cir/models_negplus.py:48-50 (element_wise_sum), :130-142 (bank_large_step)
and :150-154 (infonce_loss with nn.CrossEntropyLoss, mean reduction); the zero-shot variant
zscir/models_bank.py:90-101 differs only in which index selects the reference row and in an
optional label_smoothing (zscir/models_bank.py:35).
"""
import torch
import torch.nn.functional as F


def combine(ref_feats, text_feats):
    # clip4cir/models_negplus.py:48-50
    return ref_feats + text_feats


def l2_normalize(x, eps=1e-12):
    # F.normalize(p=2, dim=1): x / max(||x||, eps)  (clip4cir/models_negplus.py:137)
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def infonce(q, bank, labels, tau, label_smoothing=0.0):
    """clip4cir/models_negplus.py:150-154: mean_i( logsumexp_j(l_ij) - l_i,y_i ), l = q bank^T / tau."""
    logits = (q @ bank.t()) / tau
    return F.cross_entropy(logits, labels.long(), label_smoothing=label_smoothing)


def bank_large_step(refer_bank, refer_idx, text_feats, target_bank, labels, tau, label_smoothing=0.0):
    """clip4cir/models_negplus.py:130-142.  ``refer_idx`` is reference_index_all when --plus,
    else the triplet index; ``labels`` are unique-image ids (rows of target_bank)."""
    r = refer_bank[refer_idx.long()].detach()
    q = l2_normalize(combine(r, text_feats))
    return infonce(q, target_bank.detach(), labels, tau, label_smoothing)


def quantize_e4m3(bank):
    """BASELINE config 5 storage of the static bank: per-row scale = max|x| / 448 (1 for a zero row),
    bytes = round-to-nearest-even OCP e4m3fn of x / scale.  Returns (uint8 [M, D], fp32 scale [M]).  The
    reference has no fp8 path; this restates the format definition (OCP 8-bit floating point, e4m3fn: bias 7,
    max 448, no infinities) through torch's float8_e4m3fn cast."""
    bank = bank.float()
    mx = bank.abs().amax(dim=1)
    scale = torch.where(mx > 0, mx / 448.0, torch.ones_like(mx))
    q = (bank / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequantize_e4m3(data, scale):
    return data.view(torch.float8_e4m3fn).float() * scale[:, None]


def split_query_e4m3(q):
    """The query side of the fp8-MFMA bank pass (spn4cir_amd/csrc/bank.hip, bank_fp8_fwd_kernel; BASELINE config 5): the
    matrix instruction takes both operands in e4m3, so every query row is carried by TWO e4m3 terms with one scale,
        q ~= s * hi + (s / 16) * lo,   s = max|q| / 448 (1 for a zero row),   r = 1 / s,
        hi = e4m3(q * r),   lo = e4m3((q - hi * s) * (16 * r)),
    all in fp32 (every product and the difference rounded separately) with round-to-nearest-even casts.  Returns the
    fp32 value the kernel's two products represent."""
    q = q.float()
    mx = q.abs().amax(dim=1)
    s = torch.where(mx > 0, mx / 448.0, torch.ones_like(mx))[:, None]
    r = 1.0 / s
    hi = (q * r).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    lo = ((q - hi * s) * (r * 16.0)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return hi * s + lo * (s * 0.0625)


def inbatch_step(refer_feats, text_feats, target_feats, tau):
    """clip4cir/models.py:151-167 with wo_bank=True (BASELINE config 1): in-batch B x B InfoNCE between
    normalize(refer + text) and normalize(target), labels = arange(B)."""
    q = l2_normalize(combine(refer_feats, text_feats))
    t = l2_normalize(target_feats)
    return infonce(q, t, torch.arange(q.shape[0]), tau)


def infonce_stats(q, bank, labels, tau):
    """Per-row pieces used by the kernel tests: (row_lse, label_logit, row_loss)."""
    logits = (q.double() @ bank.double().t()) / tau
    lse = torch.logsumexp(logits, dim=1)
    lab = logits[torch.arange(q.shape[0]), labels.long()]
    return lse, lab, lse - lab


def infonce_grad_q(q, bank, labels, tau, grad_scale=None):
    """d(mean CE)/dq = ((softmax - onehot) / B) @ bank / tau, in fp64 for the kernel tests."""
    B = q.shape[0]
    logits = (q.double() @ bank.double().t()) / tau
    g = torch.softmax(logits, dim=1)
    g[torch.arange(B), labels.long()] -= 1.0
    scale = (1.0 / B) if grad_scale is None else grad_scale
    return (g @ bank.double()) * (scale / tau)


def tokmax_infonce(fusion_feats, target_feats, target_indexs, temp):
    """blip24cir/lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:253-265 (forward_stage2, loss_qtc):
    per sample i, sim[m, k] = <fusion_feats[i], target_feats[m, k, :]> over the 32 Q-Former tokens of every bank
    target, logit[m] = max_k sim[m, k] / temp, cross entropy against target_indexs[i]; mean over the batch.
    PINNED: tests/golden/make_golden_blip2.py runs the reference's own forward_stage2 (module loaded with import-only
    stand-ins for its LAVIS imports, the Q-Former output as a fixed input) and tests/test_oracle_golden.py checks this
    function against the captured loss and gradients (tests/golden/blip2_stage2.npz)."""
    bs = target_indexs.shape[0]
    loss = torch.zeros((), dtype=fusion_feats.dtype)
    for i in range(bs):
        sim = torch.matmul(target_feats, fusion_feats[i])          # [M, 32]
        sim_q2t, _ = sim.max(-1)
        loss = loss + F.cross_entropy((sim_q2t / temp).unsqueeze(0), target_indexs[i:i + 1].long())
    return loss / bs
