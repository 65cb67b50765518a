"""Oracle-side noise model (TEST INFRASTRUCTURE, like everything under oracle/): the same CPU restatement with every
matrix-product operand rounded to bf16, fp32 accumulation - the arithmetic class of the MI355X kernels (bf16 MFMA operands,
fp32 accumulators) and of the reference's own CUDA path (torch.cuda.amp.autocast, clip4cir/train_negplus.py:110: fp16 operands
AND fp16 outputs).  Running an oracle step under `bf16_gemm_operands()` and comparing it with the fp32 run of the same step
gives, per tensor, the error that operand rounding ALONE produces: the noise floor the GPU gradient gates are calibrated
against (tests/golden/make_noise_floor.py -> noise_floor.json; tests assert HIP error <= 1.5 x floor).

Mechanism: a TorchFunctionMode intercepts torch.matmul / Tensor.__matmul__ / F.linear / F.conv2d and replaces
`f(a, b)` by `G(f(R(a), R(b)))` where R rounds to bf16 in the forward pass (identity gradient) and G is the identity whose
backward rounds the incoming gradient to bf16 - autograd's own matmul backward then forms dA = R(g) R(b)^T and
dB = R(a)^T R(g) from rounded operands, as the backward GEMM kernels do.  Nothing in the oracle modules changes.
`outputs=True` additionally rounds every product's OUTPUT (autocast's storage class) - an upper reference, not the gate.
"""
import contextlib

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode


def _rb(x):
    return x.to(torch.bfloat16).to(x.dtype)


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


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


_MATMULS = {torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.mm, torch.bmm, torch.Tensor.mm, torch.Tensor.bmm}


class _Bf16Operands(TorchFunctionMode):
    def __init__(self, outputs=False):
        super().__init__()
        self.outputs = outputs
        self.products = 0

    def _wrap(self, y):
        self.products += 1
        return (_RoundBoth if self.outputs else _RoundBwd).apply(y)

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _MATMULS and len(args) == 2 and all(torch.is_tensor(a) and a.is_floating_point() for a in args):
            if args[0].dtype == torch.float32 and args[1].dtype == torch.float32:
                return self._wrap(func(_RoundFwd.apply(args[0]), _RoundFwd.apply(args[1]), **kwargs))
        elif func is torch.Tensor.__rmatmul__ and len(args) == 2 and all(torch.is_tensor(a) for a in args):
            if args[0].dtype == torch.float32 and args[1].dtype == torch.float32:
                return self._wrap(func(_RoundFwd.apply(args[0]), _RoundFwd.apply(args[1]), **kwargs))
        elif func in (F.linear, F.conv2d) and len(args) >= 2 and args[0].dtype == torch.float32:
            rest = args[2:]
            return self._wrap(func(_RoundFwd.apply(args[0]), _RoundFwd.apply(args[1]), *rest, **kwargs))
        return func(*args, **kwargs)


@contextlib.contextmanager
def bf16_gemm_operands(outputs=False):
    """with bf16_gemm_operands() as m: ...   every fp32 matrix product inside runs on bf16-rounded operands (forward and
    backward), fp32 accumulation; fp64 products (the oracle's reference scoring) are left alone.  m.products counts them."""
    mode = _Bf16Operands(outputs)
    with mode:
        yield mode


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
