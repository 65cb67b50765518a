"""Oracle-side noise model (TEST INFRASTRUCTURE, like everything under oracle/): the same CPU restatement with every
matrix-product operand rounded to bf16, fp32 accumulation - the arithmetic class of the MI355X kernels (bf16 MFMA operands,
fp32 accumulators) and of the reference's own CUDA path (torch.cuda.amp.autocast, clip4cir/train_negplus.py:110: fp16 operands
AND fp16 outputs).  Running an oracle step under `bf16_gemm_operands()` and comparing it with the fp32 run of the same step
gives, per tensor, the error that operand rounding ALONE produces: the noise floor the GPU gradient gates are calibrated
against (tests/golden/make_noise_floor.py -> noise_floor.json; tests assert HIP error <= 1.5 x floor).

Mechanism: a TorchFunctionMode intercepts torch.matmul / Tensor.__matmul__ / F.linear / F.conv2d and replaces
`f(a, b)` by `G(f(R(a), R(b)))` where R rounds to bf16 in the forward pass (identity gradient) and G is the identity whose
backward rounds the incoming gradient to bf16 - autograd's own matmul backward then forms dA = R(g) R(b)^T and
dB = R(a)^T R(g) from rounded operands, as the backward GEMM kernels do; a bias added to a product gets the column sums of
the same rounded R(g) (what a GEMM-based bias gradient - and autocast's fp16 grad_output.sum - reads).  Nothing in the
oracle modules changes.
`outputs=True` additionally rounds every product's OUTPUT (autocast's storage class) - an upper reference, not the gate.
"""
import contextlib

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode


_SCALE = [1.0]        # the rounding realisation of the current context (see bf16_gemm_operands(realisation=))


def _rb(x):
    s = _SCALE[0]
    if s == 1.0:
        return x.to(torch.bfloat16).to(x.dtype)
    return (x * s).to(torch.bfloat16).to(x.dtype) / s


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


class _AddBias(torch.autograd.Function):
    """a + bias (bias along dimension `dim` of a).  Backward: a gets g as it is (a residual stream's gradient stays fp32); the
    bias gets the sums of the bf16-ROUNDED g over every other dimension - the column sums a GEMM-based bias gradient forms from
    the same rounded operand its weight-gradient product reads."""

    @staticmethod
    def forward(ctx, a, b, dim):
        ctx.dim = dim % a.dim()
        shape = [1] * a.dim()
        shape[ctx.dim] = -1
        return a + b.view(shape)

    @staticmethod
    def backward(ctx, g):
        dims = [d for d in range(g.dim()) if d != ctx.dim]
        return g, _rb(g).sum(dim=dims), None


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


_ADDS = {torch.add, torch.Tensor.add, torch.Tensor.__add__, torch.Tensor.__radd__}
_MATMULS = {torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.mm, torch.bmm, torch.Tensor.mm, torch.Tensor.bmm}


class _Bf16Operands(TorchFunctionMode):
    def __init__(self, outputs=False):
        super().__init__()
        self.outputs = outputs
        self.products = 0

    def _wrap(self, y):
        self.products += 1
        y = (_RoundBoth if self.outputs else _RoundBwd).apply(y)
        y._spn_product = True            # lets the bias add that consumes it see where it came from (below)
        return y

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _MATMULS and len(args) == 2 and all(torch.is_tensor(a) and a.is_floating_point() for a in args):
            if args[0].dtype == torch.float32 and args[1].dtype == torch.float32:
                return self._wrap(func(_RoundFwd.apply(args[0]), _RoundFwd.apply(args[1]), **kwargs))
        elif func is torch.Tensor.__rmatmul__ and len(args) == 2 and all(torch.is_tensor(a) for a in args):
            if args[0].dtype == torch.float32 and args[1].dtype == torch.float32:
                return self._wrap(func(_RoundFwd.apply(args[0]), _RoundFwd.apply(args[1]), **kwargs))
        elif func in (F.linear, F.conv2d) and len(args) >= 2 and args[0].dtype == torch.float32:
            bias = args[2] if len(args) > 2 else kwargs.get("bias")
            kw = {k: v for k, v in kwargs.items() if k != "bias"}
            y = self._wrap(func(_RoundFwd.apply(args[0]), _RoundFwd.apply(args[1]), None, *args[3:], **kw))
            if bias is None:
                return y
            return _AddBias.apply(y, bias, 1 if func is F.conv2d else -1)
        elif func in _ADDS and len(args) == 2 and torch.is_tensor(args[0]) and torch.is_tensor(args[1]):
            # `product + bias` (also `(x + product) + bias`, the residual form): the bias gradient of a GEMM-based implementation is
            # the column sum of the ROUNDED output gradient - the same bf16 operand the weight-gradient product reads.  Only the
            # bias branch sees the rounding: the residual stream's own gradient stays fp32.
            a, b = args
            if getattr(b, "_spn_product", False) and not getattr(a, "_spn_product", False):
                a, b = b, a
            if getattr(a, "_spn_product", False):
                if b.dim() == 1 and b.requires_grad and b.dtype == torch.float32 and b.shape[0] == a.shape[-1] and not kwargs:
                    y = _AddBias.apply(a, b, -1)
                else:
                    y = func(*args, **kwargs)
                if torch.is_tensor(y) and y.shape == a.shape:
                    y._spn_product = True            # x + product: a bias may still follow
                return y
        return func(*args, **kwargs)


_REALISATION_SCALES = (1.0, 1.1892071, 1.4142135, 1.6817928)      # 2^(0, 1/4, 1/2, 3/4): four different mantissa alignments


@contextlib.contextmanager
def bf16_gemm_operands(outputs=False, realisation=0):
    """with bf16_gemm_operands() as m: ...   every fp32 matrix product inside runs on bf16-rounded operands (forward and
    backward), fp32 accumulation; fp64 products (the oracle's reference scoring) are left alone.  m.products counts them.
    realisation r > 0: the operands are rounded on a grid shifted by 2^(r/4) (x -> bf16(x s) / s): the same arithmetic class
    with a different, equally legitimate rounding pattern - one more sample of the noise a bf16 implementation may show.
    Gradients that are small remainders of cancelling sums (bias / LayerNorm vectors at tiny batches) vary by tens of per cent
    between such samples; the floor of a tensor is the RMS over the samples."""
    mode = _Bf16Operands(outputs)
    old = _SCALE[0]
    _SCALE[0] = _REALISATION_SCALES[realisation % len(_REALISATION_SCALES)]
    try:
        with mode:
            yield mode
    finally:
        _SCALE[0] = old


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
