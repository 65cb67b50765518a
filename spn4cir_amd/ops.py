"""torch-tensor wrappers over the C-ABI (include/spn4cir_hip.h).

PyTorch is used only for device memory and streams: every function takes CUDA(=HIP) tensors,
passes raw device pointers plus torch's current stream to libspn4cir_hip.so and returns
tensors.  There is no CPU fallback: CPU tensors are rejected.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, lib

ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF = 0, 1, 2
BANK_DIMS = (128, 256, 512, 640, 768, 1024)

_ws_cache = {}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("spn4cir_amd ops need device tensors (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


_POISON = os.environ.get("SPN_DEBUG_POISON") == "1"


def scratch_bytes(nbytes, device):
    """Uninitialised device scratch (uint8).  SPN_DEBUG_POISON=1 fills every such allocation with 0xFF bytes - NaN as fp32
    and as bf16 - so that a kernel that reads scratch it never wrote shows up as NaN in its output instead of as
    allocator-dependent noise (tests/test_poison_gpu.py runs the training steps that way)."""
    buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
    if _POISON:
        buf.fill_(255)
    return buf


def workspace(nbytes, device, slot="default"):
    """Grow-only scratch buffer per (device, slot); ops on one stream use it sequentially."""
    key = (str(device), slot)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = scratch_bytes(max(int(nbytes), 256), device)
        _ws_cache[key] = buf
    return buf


def _req(t, dtype, name):
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")


def check_index_range(idx, n, name):
    """IndexError for a row index outside [0, n), as `refer_bank[refer_indexs]` raises in the reference
    (models_negplus.py:133).  Free for host tensors - the DataLoader hands the reference's loop CPU LongTensors
    (train_negplus.py:107-111); a device tensor is only inspected (one sync) under SPN_CHECK_INDICES=1, otherwise the
    kernels' own guards apply (NaN query row / infinite loss, never an out-of-bounds read)."""
    if idx is None or idx.numel() == 0:
        return
    if idx.is_cuda and os.environ.get("SPN_CHECK_INDICES") != "1":
        return
    lo, hi = int(idx.min()), int(idx.max())
    if lo < 0 or hi >= n:
        raise IndexError(f"{name}: index {lo if lo < 0 else hi} is out of bounds for a bank with {n} rows")


# ------------------------------------------------------------------------------- GEMMs
def gemm_nt(a, b, bias=None, act=ACT_NONE, out_dtype=torch.bfloat16, want_pre=False):
    """a [M,K] bf16, b [N,K] bf16 -> act(a @ b.T + bias) ; optionally also the pre-activation."""
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b")
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    pre = torch.empty(M, N, dtype=torch.bfloat16, device=a.device) if want_pre else None
    ob, of = (out, None) if out_dtype == torch.bfloat16 else (None, out)
    check(lib().spn_gemm_nt(_p(a), _p(b), M, N, K, K, K, _p(bias), act, _p(ob), _p(of), _p(pre), N, _stream()), "gemm_nt")
    return (out, pre) if want_pre else out


def gemm_nt_resid(a, b, bias, resid):
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b"); _req(resid, torch.float32, "resid")
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    check(lib().spn_gemm_nt_resid(_p(a), _p(b), M, N, K, K, K, _p(bias), _p(resid), N, _p(out), None, N, _stream()),
          "gemm_nt_resid")
    return out


def gemm_nt_dact(a, b, pre, act):
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b"); _req(pre, torch.bfloat16, "pre")
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    check(lib().spn_gemm_nt_dact(_p(a), _p(b), M, N, K, K, K, _p(pre), act, _p(out), N, _stream()), "gemm_nt_dact")
    return out


def gemm_tn(a, b, alpha=1.0, out=None, accumulate=False, want_colsum=False):
    """a [Kr,N1] bf16, b [Kr,N2] bf16 -> a.T @ b  (fp32 [N1,N2]) [, column sums of a]."""
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b")
    Kr, N1 = a.shape
    N2 = b.shape[1]
    if out is None:
        out = torch.empty(N1, N2, dtype=torch.float32, device=a.device)
    nb = lib().spn_gemm_tn_workspace_bytes(Kr, N1, N2)
    ws = workspace(nb, a.device)
    cs = torch.empty(N1, dtype=torch.float32, device=a.device) if want_colsum else None
    check(lib().spn_gemm_tn(_p(a), _p(b), Kr, N1, N2, N1, N2, _p(out), N2, float(alpha), int(accumulate), _p(cs),
                            _p(ws), ws.numel(), _stream()), "gemm_tn")
    return (out, cs) if want_colsum else out


def gemm_tn_pair(a1, b1, a2, b2):
    """(a1.T @ b1, colsum(a1), a2.T @ b2, colsum(a2)) for bf16 [Kr, N] operands sharing Kr, in one launch."""
    for t in (a1, b1, a2, b2):
        _req(t, torch.bfloat16, "operand")
    Kr = a1.shape[0]
    N1a, N2a, N1b, N2b = a1.shape[1], b1.shape[1], a2.shape[1], b2.shape[1]
    dev = a1.device
    c1 = torch.empty(N1a, N2a, dtype=torch.float32, device=dev); s1 = torch.empty(N1a, dtype=torch.float32, device=dev)
    c2 = torch.empty(N1b, N2b, dtype=torch.float32, device=dev); s2 = torch.empty(N1b, dtype=torch.float32, device=dev)
    ws = workspace(lib().spn_gemm_tn_pair_workspace_bytes(Kr, N1a, N2a, N1b, N2b), dev)
    check(lib().spn_gemm_tn_pair(_p(a1), _p(b1), N1a, N2a, _p(c1), _p(s1), _p(a2), _p(b2), N1b, N2b, _p(c2), _p(s2), Kr,
                                 _p(ws), ws.numel(), _stream()), "gemm_tn_pair")
    return c1, s1, c2, s2


class _TnProblem(C.Structure):     # == spn_tn_problem
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("colsum", C.c_void_p),
                ("N1", C.c_int), ("N2", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int)]


def gemm_tn_grouped(pairs, want_colsum=True):
    """[(a [Kr,N1] bf16, b [Kr,N2] bf16), ...] sharing Kr -> [(a.T @ b fp32 [N1,N2], colsum(a) [N1] or None), ...]
    in ONE launch without split-K (spn_gemm_tn_grouped)."""
    Kr = pairs[0][0].shape[0]
    dev = pairs[0][0].device
    arr = (_TnProblem * len(pairs))()
    outs = []
    for i, (a, b) in enumerate(pairs):
        _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b")
        if a.shape[0] != Kr or b.shape[0] != Kr:
            raise ValueError("grouped TN problems must share the reduction length")
        N1, N2 = a.shape[1], b.shape[1]
        c = torch.empty(N1, N2, dtype=torch.float32, device=dev)
        cs = torch.empty(N1, dtype=torch.float32, device=dev) if want_colsum else None
        arr[i] = _TnProblem(a.data_ptr(), b.data_ptr(), c.data_ptr(), cs.data_ptr() if want_colsum else None, N1, N2, N1, N2, N2)
        outs.append((c, cs))
    ws = workspace(lib().spn_gemm_tn_grouped_workspace_bytes(Kr), dev)
    check(lib().spn_gemm_tn_grouped(C.byref(arr), len(pairs), Kr, _p(ws), ws.numel(), _stream()), "gemm_tn_grouped")
    return outs


# ------------------------------------------------------------------------- elementwise
def cast_bf16(x):
    _req(x, torch.float32, "x")
    y = torch.empty_like(x, dtype=torch.bfloat16)
    check(lib().spn_cast_f32_bf16(_p(x), _p(y), x.numel(), _stream()), "cast")
    return y


def cast_transpose_bf16(x):
    _req(x, torch.float32, "x")
    r, c = x.shape
    y = torch.empty(r, c, dtype=torch.bfloat16, device=x.device)
    yt = torch.empty(c, r, dtype=torch.bfloat16, device=x.device)
    check(lib().spn_cast_transpose_f32_bf16(_p(x), _p(y), _p(yt), r, c, _stream()), "cast_transpose")
    return y, yt


def colsum(x):
    _req(x, torch.bfloat16, "x")
    r, c = x.shape
    out = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = workspace(lib().spn_colsum_workspace_bytes(r, c), x.device)
    check(lib().spn_colsum_bf16(_p(x), r, c, c, _p(out), 0, _p(ws), ws.numel(), _stream()), "colsum")
    return out


# --------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, eps=1e-5, out_dtype=torch.bfloat16):
    _req(x, torch.float32, "x")
    rows, W = x.shape
    y = torch.empty(rows, W, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    yb, yf = (y, None) if out_dtype == torch.bfloat16 else (None, y)
    check(lib().spn_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(yb), _p(yf), _p(mean), _p(rstd), rows, W, eps,
                                  _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_accum=None):
    """Returns (dx fp32, dx bf16, dgamma, dbeta); dx_accum (fp32) is added into if given."""
    rows, W = x.shape
    dx = dx_accum if dx_accum is not None else torch.empty(rows, W, dtype=torch.float32, device=x.device)
    dxb = torch.empty(rows, W, dtype=torch.bfloat16, device=x.device)
    dg = torch.empty(W, dtype=torch.float32, device=x.device)
    db = torch.empty(W, dtype=torch.float32, device=x.device)
    ws = workspace(lib().spn_layernorm_bwd_workspace_bytes(rows, W), x.device)
    dyb, dyf = (dy, None) if dy.dtype == torch.bfloat16 else (None, dy)
    check(lib().spn_layernorm_bwd(_p(dyb), _p(dyf), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx),
                                  int(dx_accum is not None), _p(dxb), _p(dg), _p(db), 0, rows, W, _p(ws), ws.numel(),
                                  _stream()), "layernorm_bwd")
    return dx, dxb, dg, db


# --------------------------------------------------------------------------- attention
def attention_fwd(q, k, v, B, H, Lq, Lk, causal=False, key_bias=None, scale=0.125):
    """q [B*Lq, H*64], k/v [B*Lk, H*64] bf16 (may be column views of a packed qkv) -> (o, lse)."""
    o = torch.empty(B * Lq, H * 64, dtype=torch.bfloat16, device=q.device)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
    check(lib().spn_attention_fwd(_p(q), _p(k), _p(v), q.stride(0), k.stride(0), v.stride(0), _p(o), H * 64, _p(lse),
                                  _p(key_bias), B, H, Lq, Lk, int(causal), scale, _stream()), "attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, lse, d_o, B, H, Lq, Lk, causal=False, key_bias=None, scale=0.125):
    dq = torch.empty(B * Lq, H * 64, dtype=torch.bfloat16, device=q.device)
    dk = torch.empty(B * Lk, H * 64, dtype=torch.bfloat16, device=q.device)
    dv = torch.empty(B * Lk, H * 64, dtype=torch.bfloat16, device=q.device)
    delta = torch.empty(B, H, Lq, dtype=torch.float32, device=q.device)
    check(lib().spn_attention_bwd(_p(q), _p(k), _p(v), q.stride(0), k.stride(0), v.stride(0), _p(o), o.stride(0),
                                  _p(lse), _p(key_bias), _p(d_o), d_o.stride(0), _p(dq), _p(dk), _p(dv), H * 64, H * 64,
                                  H * 64, _p(delta), B, H, Lq, Lk, int(causal), scale, _stream()), "attention_bwd")
    return dq, dk, dv


# ------------------------------------------------------------------- combiner + bank loss
def bank_dim(D):
    """Smallest supported bank width >= D (queries/banks are zero-padded to it)."""
    for d in BANK_DIMS:
        if d >= D:
            return d
    raise ValueError(f"embedding dim {D} > {BANK_DIMS[-1]} not supported by the bank kernels")


class Fp8Bank:
    """Static bank as OCP e4m3 bytes [M, Dp] + one fp32 scale per row (BASELINE config 5).  Accepted wherever a
    bf16 bank is (bank_stats_fwd / bank_grad_q dispatch on it)."""

    def __init__(self, data, scale):
        self.data, self.scale = data, scale

    @property
    def shape(self):
        return self.data.shape

    @property
    def device(self):
        return self.data.device

    def dequantize(self):
        """fp32 [M, Dp] exactly as the kernels see it before the bf16 rounding."""
        return self.data.view(torch.float8_e4m3fn).float() * self.scale[:, None]

    def bf16_image(self):
        """bf16 [M, Dp] = the expansion the fp8 kernels make per pass at large batches, made ONCE and kept
        (spn_bank_dequant_fp8).  Used by the bank calls from FP8_IMAGE_MIN_B queries per call on."""
        if getattr(self, "_image", None) is None:
            M, Dp = self.data.shape
            img = torch.empty(M, Dp, dtype=torch.bfloat16, device=self.data.device)
            check(lib().spn_bank_dequant_fp8(_p(self.data), _p(self.scale), M, Dp, _p(img), _stream()), "bank_dequant_fp8")
            self._image = img
        return self._image


# From this many queries per call on the bank pass is bound by MFMA issue, not by the bank's bytes: an e4m3 bank is used through
# its kept bf16 image (same values as the kernels' per-pass expansion; +2 B per element of HBM) and the saved-probabilities
# backward applies.  Below it the e4m3 bytes are streamed (half the bytes of the one HBM stream of the loss).
FP8_IMAGE_MIN_B = 256


def fp8_image_min_b():
    """Query count from which an e4m3 bank is served through its kept bf16 image.  SPN_FP8_IMAGE_MIN_B overrides the default
    (256): 0 = never build the image (memory-bound deployments: +2 bytes per bank element stay unspent; large batches then
    expand the shard per pass inside the library), any other value = the threshold."""
    v = os.environ.get("SPN_FP8_IMAGE_MIN_B")
    if v is None or v.strip() == "":
        return FP8_IMAGE_MIN_B
    n = int(v)
    return (1 << 62) if n <= 0 else n


def bank_operand_kind(bank, B):
    """What the bank calls actually read for `B` queries per call: "bf16", "e4m3" (raw bytes on the fp8 MFMA) or "bf16_image"
    (an e4m3 bank through its kept expansion) - reported by bench.py next to every fp8 number."""
    if isinstance(bank, Fp8Bank):
        return "bf16_image" if B >= fp8_image_min_b() else "e4m3"
    return "bf16"


def _bank_operand(bank, B):
    if isinstance(bank, Fp8Bank) and B >= fp8_image_min_b():
        return bank.bf16_image()
    return bank


def prepare_bank(bank_f32, dtype="bf16"):
    """fp32 [M, D] (L2-normalised rows) -> device bf16 [M, bank_dim(D)], zero padded; dtype="fp8" -> Fp8Bank."""
    M, D = bank_f32.shape
    Dp = bank_dim(D)
    if dtype == "fp8":
        _req(bank_f32, torch.float32, "bank")
        data = torch.empty(M, Dp, dtype=torch.uint8, device=bank_f32.device)
        scale = torch.empty(M, dtype=torch.float32, device=bank_f32.device)
        check(lib().spn_bank_quantize_fp8(_p(bank_f32.contiguous()), M, D, Dp, _p(data), _p(scale), _stream()),
              "bank_quantize_fp8")
        return Fp8Bank(data, scale)
    if dtype != "bf16":
        raise ValueError(dtype)
    out = torch.zeros(M, Dp, dtype=torch.bfloat16, device=bank_f32.device)
    out[:, :D] = bank_f32.to(torch.bfloat16)
    return out


def combine_l2norm_fwd(refer_bank, ref_idx, text, ldq=None):
    """-> (q fp32 [B,D], q bf16 [B,ldq], inv_norm [B]).  A ref_idx outside the bank gives a NaN row (the kernel never
    dereferences it); callers that still hold the indices on the host raise IndexError first (check_index_range)."""
    _req(text, torch.float32, "text")
    B, D = text.shape
    n_refer = 0
    if refer_bank is not None:
        _req(refer_bank, torch.float32, "refer_bank")
        _req(ref_idx, torch.int64, "ref_idx")
        if refer_bank.dim() != 2 or refer_bank.shape[1] != D or ref_idx.numel() != B:
            raise ValueError(f"refer_bank {tuple(refer_bank.shape)} / ref_idx {tuple(ref_idx.shape)} do not match text {tuple(text.shape)}")
        n_refer = refer_bank.shape[0]
    ldq = ldq or bank_dim(D)
    q = torch.empty(B, D, dtype=torch.float32, device=text.device)
    qb = torch.empty(B, ldq, dtype=torch.bfloat16, device=text.device)
    inv = torch.empty(B, dtype=torch.float32, device=text.device)
    check(lib().spn_combine_l2norm_fwd(_p(refer_bank), _p(ref_idx), n_refer, _p(text), _p(q), _p(qb), _p(inv), B, D, ldq,
                                       _stream()), "combine_l2norm_fwd")
    return q, qb, inv


def combine_l2norm_bwd(q, inv_norm, dq, scale=None):
    """scale: optional 1-element fp32 device tensor multiplied into the result on the device (autograd's incoming
    d(loss): no host synchronisation on it)."""
    B, D = q.shape
    dtext = torch.empty(B, D, dtype=torch.float32, device=q.device)
    if scale is not None:
        if scale.dtype != torch.float32 or not scale.is_cuda or scale.numel() != 1:
            raise ValueError("scale must be a 1-element fp32 device tensor")
        check(lib().spn_combine_l2norm_bwd_scaled(_p(q), _p(inv_norm), _p(dq), _p(scale), _p(dtext), B, D, _stream()),
              "combine_l2norm_bwd_scaled")
        return dtext
    check(lib().spn_combine_l2norm_bwd(_p(q), _p(inv_norm), _p(dq), _p(dtext), B, D, _stream()), "combine_l2norm_bwd")
    return dtext


BANK_LOGITS_MAX_BYTES = 1 << 30


def bank_logits_buffer(B, M, device):
    """Scratch of the forward/backward pair (bank_stats_fwd(..., save=buf) / bank_grad_q(..., saved=buf)): the chunk
    partials of the fused single pass, or the saved probabilities / logits of the two-pass pairs - the library routes by
    shape.  None when it would exceed BANK_LOGITS_MAX_BYTES (the backward pass then recomputes the logits).  One buffer
    per forward call (torch's caching allocator): it belongs to that call's autograd / step context until its backward
    has run, so two forwards in flight never share it."""
    n = lib().spn_bank_logits_bytes(B, M)
    if n > BANK_LOGITS_MAX_BYTES:
        return None
    return scratch_bytes(n, device)


def bank_stats_fwd(q_bf16, bank_bf16, labels, inv_tau, m_begin=0, save=None):
    """save (bank_logits_buffer): keep the logits of this call for bank_grad_q(..., saved=save) - no recomputation in the
    backward pass (spn_bank_stats_fwd_save)."""
    B, Dp = q_bf16.shape
    M = bank_bf16.shape[0]
    bank_bf16 = _bank_operand(bank_bf16, B)
    stats = torch.empty(B, 4, dtype=torch.float32, device=q_bf16.device)
    fp8 = isinstance(bank_bf16, Fp8Bank)
    ws = workspace((lib().spn_bank_workspace_bytes_fp8 if fp8 else lib().spn_bank_workspace_bytes)(B, M, Dp), q_bf16.device,
                   "bank")
    if save is not None:
        if not save.is_cuda or save.numel() * save.element_size() < lib().spn_bank_logits_bytes(B, M):
            raise ValueError("save: device scratch of spn_bank_logits_bytes(B, M) bytes (ops.bank_logits_buffer)")
        data, scale = (bank_bf16.data, bank_bf16.scale) if fp8 else (bank_bf16, None)
        check(lib().spn_bank_stats_fwd_save(_p(q_bf16), Dp, _p(data), _p(scale), _p(labels), B, M, Dp, m_begin, inv_tau,
                                            _p(stats), _p(save), _p(ws), ws.numel(), _stream()), "bank_stats_fwd_save")
        return stats
    if fp8:
        check(lib().spn_bank_stats_fwd_fp8(_p(q_bf16), Dp, _p(bank_bf16.data), _p(bank_bf16.scale), _p(labels), B, M, Dp,
                                           m_begin, inv_tau, _p(stats), _p(ws), ws.numel(), _stream()), "bank_stats_fwd_fp8")
        return stats
    check(lib().spn_bank_stats_fwd(_p(q_bf16), Dp, _p(bank_bf16), _p(labels), B, M, Dp, m_begin, inv_tau, _p(stats),
                                   _p(ws), ws.numel(), _stream()), "bank_stats_fwd")
    return stats


def bank_loss_finalize(stats, M_total, label_smoothing=0.0):
    """stats [nshards, B, 4] or [B, 4] -> (row_lse [B], row_loss [B], loss_mean [1])"""
    if stats.dim() == 2:
        stats = stats.unsqueeze(0)
    stats = stats.contiguous()
    n, B, _ = stats.shape
    lse = torch.empty(B, dtype=torch.float32, device=stats.device)
    row = torch.empty(B, dtype=torch.float32, device=stats.device)
    mean = torch.empty(1, dtype=torch.float32, device=stats.device)
    check(lib().spn_bank_loss_finalize(_p(stats), n, B, M_total, label_smoothing, _p(lse), _p(row), _p(mean), _stream()),
          "bank_loss_finalize")
    return lse, row, mean


def bank_grad_q(q_bf16, bank_bf16, labels, inv_tau, row_lse, grad_scale, M_total=None, label_smoothing=0.0, m_begin=0,
                saved=None):
    """saved: the buffer the matching bank_stats_fwd(..., save=...) call filled (spn_bank_grad_q_saved)."""
    B, Dp = q_bf16.shape
    M = bank_bf16.shape[0]
    bank_bf16 = _bank_operand(bank_bf16, B)
    dq = torch.empty(B, Dp, dtype=torch.float32, device=q_bf16.device)
    fp8 = isinstance(bank_bf16, Fp8Bank)
    ws = workspace((lib().spn_bank_workspace_bytes_fp8 if fp8 else lib().spn_bank_workspace_bytes)(B, M, Dp), q_bf16.device,
                   "bank")
    if saved is not None:
        data, scale = (bank_bf16.data, bank_bf16.scale) if fp8 else (bank_bf16, None)
        check(lib().spn_bank_grad_q_saved(_p(q_bf16), Dp, _p(data), _p(scale), _p(labels), B, M, Dp, m_begin, inv_tau,
                                          _p(saved), _p(row_lse), label_smoothing, M_total or M, grad_scale, _p(dq), _p(ws),
                                          ws.numel(), _stream()), "bank_grad_q_saved")
        return dq
    if fp8:
        check(lib().spn_bank_grad_q_fp8(_p(q_bf16), Dp, _p(bank_bf16.data), _p(bank_bf16.scale), _p(labels), B, M, Dp,
                                        m_begin, inv_tau, _p(row_lse), label_smoothing, M_total or M, grad_scale, _p(dq),
                                        _p(ws), ws.numel(), _stream()), "bank_grad_q_fp8")
        return dq
    check(lib().spn_bank_grad_q(_p(q_bf16), Dp, _p(bank_bf16), _p(labels), B, M, Dp, m_begin, inv_tau, _p(row_lse),
                                label_smoothing, M_total or M, grad_scale, _p(dq), _p(ws), ws.numel(), _stream()),
          "bank_grad_q")
    return dq


# -------------------------------------------------------------------------------- AdamW
def tau_grad(q, dqk, tau_dev, dtau_out=None, alpha=1.0, scale_dev=None):
    """-> inv_tau (1-element device tensor); dtau_out (optional 1-element fp32 device tensor) <- -alpha * scale_dev *
    (sum q . dqk) / tau^2 (spn_tau_grad; dqk may carry padding columns: its leading dimension is passed along)."""
    _req(q, torch.float32, "q")
    _req(dqk, torch.float32, "dqk")
    B, D = q.shape
    inv = torch.empty(1, dtype=torch.float32, device=q.device)
    check(lib().spn_tau_grad(_p(q), _p(dqk), dqk.stride(0), _p(tau_dev), B, D, float(alpha), _p(scale_dev), _p(dtau_out), _p(inv),
                             _stream()), "tau_grad")
    return inv


def xattn_fwd(q, wkv, bkv, x, heads, cu=None, L=None, scale=0.125, wkv_t=None):
    """Cross-attention over frozen tokens in the absorbed form (spn_xattn_fwd; blip4cir/med.py:97-181, 196-234): q bf16 [rows, W],
    wkv bf16 [2W, E] (key rows, then value rows), bkv fp32 [2W], x bf16 [B, S, E] -> (ctx bf16 [rows, W], saved) where `saved` goes
    to xattn_bwd.  Dense rows = B * L (cu None) or packed (cu int32 [B + 1] device prefix sums, L = the longest length)."""
    _req(q, torch.bfloat16, "q")
    _req(wkv, torch.bfloat16, "wkv")
    _req(x, torch.bfloat16, "x")
    _req(bkv, torch.float32, "bkv")
    B, S, E = x.shape
    rows, W = q.shape
    H = int(heads)
    if cu is None:
        if rows % B:
            raise ValueError("dense rows must be B * L")
        L = rows // B
    elif L is None:
        raise ValueError("packed rows need L (the longest caption)")
    if W != H * 64 or tuple(wkv.shape) != (2 * W, E) or not lib().spn_xattn_ok(B, L, H, S, E):
        raise ValueError("shape not supported by the absorbed cross-attention (spn_xattn_ok)")
    wkv_t = wkv.t().contiguous() if wkv_t is None else wkv_t
    SP = lib().spn_xattn_sp(S)
    qa = torch.empty(rows, H, E, dtype=torch.bfloat16, device=q.device)
    oa = torch.empty_like(qa)
    p = torch.empty(rows * H, SP, dtype=torch.bfloat16, device=q.device)
    ctx = torch.empty(rows, W, dtype=torch.bfloat16, device=q.device)
    check(lib().spn_xattn_fwd(_p(q), _p(wkv), _p(wkv_t), _p(bkv), _p(x), _p(cu), _p(qa), _p(p), _p(oa), _p(ctx), B, L, H, S, E,
                              rows if cu is not None else 0, float(scale), _stream()), "xattn_fwd")
    return ctx, dict(q=q, wkv=wkv, wkv_t=wkv_t, bkv=bkv, x=x, cu=cu, qa=qa, p=p, oa=oa, ctx=ctx, B=B, L=L, H=H, S=S, E=E, rows=rows,
                     scale=float(scale))


def xattn_bwd(saved, dctx):
    """Backward of xattn_fwd: dctx bf16 [rows, W] -> (dq bf16 [rows, W], dwkv fp32 [2W, E], dbkv fp32 [2W])."""
    _req(dctx, torch.bfloat16, "dctx")
    s = saved
    dev = dctx.device
    W = s["H"] * 64
    doa, dqa, ds = torch.empty_like(s["qa"]), torch.empty_like(s["qa"]), torch.empty_like(s["p"])
    delta = torch.empty(s["rows"] * s["H"], dtype=torch.float32, device=dev)
    dq = torch.empty(s["rows"], W, dtype=torch.bfloat16, device=dev)
    dwkv = torch.empty(2 * W, s["E"], dtype=torch.float32, device=dev)
    dbkv = torch.empty(2 * W, dtype=torch.float32, device=dev)
    check(lib().spn_xattn_bwd(_p(dctx), _p(s["ctx"]), _p(s["q"]), _p(s["wkv"]), _p(s["wkv_t"]), _p(s["bkv"]), _p(s["x"]), _p(s["cu"]),
                              _p(s["p"]), _p(s["oa"]), _p(doa), _p(ds), _p(dqa), _p(delta), _p(dq), _p(dwkv), _p(dbkv), s["B"], s["L"],
                              s["H"], s["S"], s["E"], s["rows"] if s["cu"] is not None else 0, s["scale"], _stream()), "xattn_bwd")
    return dq, dwkv, dbkv


def token_bank_bf16(bank, device, chunk=512):
    """fp32 / bf16 [N, S, E] token bank (host or device) -> contiguous bf16 [N, S, E] on `device`, uploaded and converted in
    chunks of `chunk` images (a 30 000 x 577 x 768 bank is 53 GB in fp32: never two copies of it anywhere)."""
    if bank.dim() != 3:
        raise ValueError("token bank must be [N, S, E]")
    if bank.dtype == torch.bfloat16 and bank.is_cuda and bank.is_contiguous():
        return bank
    N, S, E = bank.shape
    out = torch.empty(N, S, E, dtype=torch.bfloat16, device=device)
    for s in range(0, N, chunk):
        blk = bank[s:s + chunk].to(device, non_blocking=False)
        if blk.dtype == torch.float32:
            check(lib().spn_cast_f32_bf16(_p(blk.contiguous()), _p(out[s:s + chunk]), blk.numel(), _stream()), "cast_f32_bf16")
        else:
            out[s:s + chunk].copy_(blk)
    return out


def gather_bank_rows_bf16(bank, idx):
    """out[b] = bank[idx[b]] for a contiguous bf16 [N, ...] device bank and int64 [B] device indices (spn_gather_bank_rows_bf16)."""
    row = bank[0].numel()
    out = torch.empty((idx.numel(),) + tuple(bank.shape[1:]), dtype=torch.bfloat16, device=bank.device)
    check(lib().spn_gather_bank_rows_bf16(_p(bank), bank.shape[0], _p(idx), _p(out), idx.numel(), row, _stream()), "gather_bank_rows")
    return out


def scale_cast_bf16(x, scale_dev, reciprocal=False, ldo=None):
    """bf16(x * s) with s = the 1-element fp32 DEVICE tensor scale_dev (or 1 / s), zero padded to ldo columns: a query
    scaled by a learnable temperature without reading it on the host (spn_scale_cast_bf16)."""
    _req(x, torch.float32, "x")
    if scale_dev.dtype != torch.float32 or not scale_dev.is_cuda or scale_dev.numel() != 1:
        raise ValueError("scale_dev must be a 1-element fp32 device tensor")
    B, D = x.shape
    ldo = ldo or bank_dim(D)
    out = torch.empty(B, ldo, dtype=torch.bfloat16, device=x.device)
    check(lib().spn_scale_cast_bf16(_p(x.contiguous()), _p(scale_dev), 1 if reciprocal else 0, _p(out), B, D, ldo, _stream()),
          "scale_cast_bf16")
    return out


def negtype_head(refer, text, target, tau, neg_type):
    """clip4cir/models_negtype.py's four in-batch terms at the feature level (spn_negtype_head): fp32 [B, D] raw reference,
    text and target features -> (loss [1], d_refer, d_text, d_target)."""
    for n, x in (("refer", refer), ("text", text), ("target", target)):
        _req(x, torch.float32, n)
    B, D = refer.shape
    if text.shape != (B, D) or target.shape != (B, D):
        raise ValueError("refer / text / target must share one [B, D] shape")
    dev = refer.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dr, dt, di = (torch.empty(B, D, dtype=torch.float32, device=dev) for _ in range(3))
    ws = workspace(lib().spn_negtype_workspace_bytes(B, D), dev, "negtype")
    check(lib().spn_negtype_head(_p(refer), _p(text), _p(target), B, D, 1.0 / tau, int(neg_type), _p(loss), _p(dr), _p(dt), _p(di),
                                 _p(ws), ws.numel(), _stream()), "negtype_head")
    return loss, dr, dt, di


def bank_step_ok(B, M, Dp, bank):
    """True when bank_step serves this shape (the single-pass kernels: see spn_bank_step_ok)."""
    return bool(lib().spn_bank_step_ok(B, M, Dp, 1 if isinstance(_bank_operand(bank, B), Fp8Bank) else 0))


def bank_step(q_bf16, bank_bf16, labels, inv_tau, grad_scale, save):
    """Forward and backward of the bank loss for a bank held entirely by this process, label smoothing 0, in two launches
    (spn_bank_step): -> (row_lse [B], row_loss [B], loss_mean [1], dq [B, Dp] fp32 = grad_scale * d(sum row_loss)/dq)."""
    B, Dp = q_bf16.shape
    M = bank_bf16.shape[0]
    dev = q_bf16.device
    bank_bf16 = _bank_operand(bank_bf16, B)
    fp8 = isinstance(bank_bf16, Fp8Bank)
    data, scale = (bank_bf16.data, bank_bf16.scale) if fp8 else (bank_bf16, None)
    if save is None or not save.is_cuda or save.numel() * save.element_size() < lib().spn_bank_logits_bytes(B, M):
        raise ValueError("save: device scratch of spn_bank_logits_bytes(B, M) bytes (ops.bank_logits_buffer)")
    lse = torch.empty(B, dtype=torch.float32, device=dev)
    row = torch.empty(B, dtype=torch.float32, device=dev)
    mean = torch.empty(1, dtype=torch.float32, device=dev)
    dq = torch.empty(B, Dp, dtype=torch.float32, device=dev)
    check(lib().spn_bank_step(_p(q_bf16), Dp, _p(data), _p(scale), _p(labels), B, M, Dp, inv_tau, grad_scale, _p(save), _p(lse),
                              _p(row), _p(mean), _p(dq), _stream()), "bank_step")
    return lse, row, mean, dq


def bank_stats_fwd_tokmax(q_bf16, bank_tok_bf16, labels, inv_tau, t_begin=0):
    """Token-max bank (spn_bank_stats_fwd_tokmax): bank_tok_bf16 [n_targets, 32, Dp] bf16, labels = target ids."""
    B, Dp = q_bf16.shape
    n, g, Db = bank_tok_bf16.shape
    if g != 32 or Db != Dp or not bank_tok_bf16.is_contiguous():
        raise ValueError("token-max bank must be contiguous [n_targets, 32, %d]" % Dp)
    stats = torch.empty(B, 4, dtype=torch.float32, device=q_bf16.device)
    ws = workspace(lib().spn_bank_workspace_bytes(B, n * 32, Dp), q_bf16.device, "bank")
    check(lib().spn_bank_stats_fwd_tokmax(_p(q_bf16), Dp, _p(bank_tok_bf16), _p(labels), B, n, Dp, t_begin, inv_tau,
                                          _p(stats), _p(ws), ws.numel(), _stream()), "bank_stats_fwd_tokmax")
    return stats


def bank_grad_q_tokmax(q_bf16, bank_tok_bf16, labels, inv_tau, row_lse, grad_scale, targets_total=None,
                       label_smoothing=0.0, t_begin=0):
    B, Dp = q_bf16.shape
    n = bank_tok_bf16.shape[0]
    dq = torch.empty(B, Dp, dtype=torch.float32, device=q_bf16.device)
    ws = workspace(lib().spn_bank_workspace_bytes(B, n * 32, Dp), q_bf16.device, "bank")
    check(lib().spn_bank_grad_q_tokmax(_p(q_bf16), Dp, _p(bank_tok_bf16), _p(labels), B, n, Dp, t_begin, inv_tau,
                                       _p(row_lse), label_smoothing, targets_total or n, grad_scale, _p(dq), _p(ws),
                                       ws.numel(), _stream()), "bank_grad_q_tokmax")
    return dq


def adamw_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01, inv_scale=1.0, found_inf=None):
    check(lib().spn_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step,
                               inv_scale, _p(found_inf), _stream()), "adamw_step")


def adamw_tick(step_dev, found_inf=None):
    """step_dev (1-element fp32 device counter) += 1 unless *found_inf != 0; call once per optimizer step."""
    check(lib().spn_adamw_tick(_p(step_dev), _p(found_inf), _stream()), "adamw_tick")


def adamw_step_dev(p, g, m, v, step_dev, lr, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01, grad_scale=None, found_inf=None):
    """adamw_step with the bias-correction step read from the device counter (skipped steps do not count)."""
    check(lib().spn_adamw_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                   _p(step_dev), _p(grad_scale), _p(found_inf), _stream()), "adamw_step_dev")


def grad_check_finite(g, found_inf):
    check(lib().spn_grad_check_finite(_p(g), g.numel(), _p(found_inf), _stream()), "grad_check_finite")


# ----------------------------------------------------------------------------- Recall@K
def cosine_scores_f64(q, gallery):
    _req(q, torch.float32, "q"); _req(gallery, torch.float32, "gallery")
    Nq, D = q.shape
    Ng = gallery.shape[0]
    out = torch.empty(Nq, Ng, dtype=torch.float64, device=q.device)
    check(lib().spn_cosine_scores_f64(_p(q), _p(gallery), Nq, Ng, D, _p(out), _stream()), "cosine_scores_f64")
    return out


def topk_from_scores(scores, K, exclude=None):
    Nq, Ng = scores.shape
    idx = torch.empty(Nq, K, dtype=torch.int32, device=scores.device)
    val = torch.empty(Nq, K, dtype=torch.float64, device=scores.device)
    check(lib().spn_topk_from_scores(_p(scores), Nq, Ng, K, _p(exclude), _p(idx), _p(val), _stream()), "topk")
    return idx, val


def inbatch_grad_t(qb, tb, lse, inv_tau, grad_scale, D):
    """In-batch negatives (clip4cir/models.py:160-167): target-side gradient fp32 [B, D] of the mean CE whose
    query-side gradient is bank_grad_q(qb, tb, arange(B), ...)."""
    _req(qb, torch.bfloat16, "qb"); _req(tb, torch.bfloat16, "tb"); _req(lse, torch.float32, "lse")
    B, ld = qb.shape
    dt = torch.empty(B, D, dtype=torch.float32, device=qb.device)
    check(lib().spn_inbatch_grad_t(_p(qb), _p(tb), ld, _p(lse), B, D, float(inv_tau), float(grad_scale), _p(dt), _stream()),
          "inbatch_grad_t")
    return dt
