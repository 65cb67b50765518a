"""Token-max bank loss of the BLIP-2 / Q-Former second stage.

Reference: blip24cir/lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:226-268 (forward_stage2):
fusion_feats [B, 256] (normalised text_proj_q output) is scored against a static bank target_feats [M, 32, 256]
(32 Q-Former tokens per target image), the logit of a target is the MAX over its 32 tokens divided by the
learnable temperature, and the loss is the batch mean of the cross entropy.  The reference runs a Python loop over
the batch with one [M, 32] matmul per sample; here the bank stays on the device in bf16 and both passes stream it
once (spn_bank_stats_fwd_tokmax / spn_bank_grad_q_tokmax).  Only the loss head is built - the Q-Former itself
(LAVIS) is out of scope, so the query producer is whatever differentiable torch graph ends in fusion_feats.
"""
import torch

from . import ops


def prepare_token_bank(target_feats):
    """[M, 32, D] fp32/fp16/bf16 (any device) -> contiguous bf16 [M, 32, Dp] on the GPU (Dp = kernel width >= D)."""
    if target_feats.dim() != 3 or target_feats.shape[1] != 32:
        raise ValueError("token bank must be [M, 32, D] (32 query tokens per target)")
    M, _, D = target_feats.shape
    Dp = ops.bank_dim(D)
    out = torch.zeros(M, 32, Dp, dtype=torch.bfloat16, device="cuda")
    out[:, :, :D] = target_feats.to("cuda")
    return out


_MAX_SHARD_BYTES = (1 << 32) - 1


def _shards(bank, fn):
    """One kernel call addresses < 4 GiB of bank (32-bit buffer offsets): larger banks go through in target shards,
    exactly as ranks would hold them (statistics merged by bank_loss_finalize, dq partials summed)."""
    M, _, Dp = bank.shape
    per = max(1, _MAX_SHARD_BYTES // (32 * Dp * 2) - 1)
    return [fn(bank[t0:t0 + per], t0) for t0 in range(0, M, per)]


class _TokMaxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fusion_feats, temp, bank, labels):
        B, D = fusion_feats.shape
        Dp = bank.shape[2]
        tau = float(temp.detach()) if torch.is_tensor(temp) else float(temp)
        # the reference feeds already-normalised features; only the bf16 cast + padding happen here
        qb = torch.zeros(B, Dp, dtype=torch.bfloat16, device=fusion_feats.device)
        qb[:, :D] = fusion_feats.detach()
        stats = _shards(bank, lambda sh, t0: ops.bank_stats_fwd_tokmax(qb, sh, labels, 1.0 / tau, t_begin=t0))
        lse, row, mean = ops.bank_loss_finalize(torch.stack(stats), bank.shape[0])
        ctx.st = dict(qb=qb, bank=bank, labels=labels, lse=lse, tau=tau, B=B, D=D, q=fusion_feats.detach(),
                      temp_is_tensor=torch.is_tensor(temp))
        return mean.reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        st = ctx.st
        M = st["bank"].shape[0]
        parts = _shards(st["bank"], lambda sh, t0: ops.bank_grad_q_tokmax(st["qb"], sh, st["labels"], 1.0 / st["tau"], st["lse"],
                                                                           1.0 / st["B"], targets_total=M, t_begin=t0))
        dq = parts[0] if len(parts) == 1 else torch.stack(parts).sum(0)
        dq = dq[:, :st["D"]] * grad_out                      # the incoming d(loss) stays on the device (no host sync)
        # logits = s / temp  =>  dL/dtemp = -(1/temp) * sum_b <q_b, dL/dq_b>   (the max picks rows, it has no scale)
        dtemp = (-(st["q"] * dq).sum() / st["tau"]).reshape(()) if st["temp_is_tensor"] else None
        return dq, dtemp, None, None


def loss_qtc(fusion_feats, target_bank_bf16, target_indexs, temp):
    """-> {'loss_qtc': 0-dim tensor with grad} (the dict the reference's forward_stage2 returns, :266-268).
    fusion_feats: CUDA fp32 [B, D], normalised; target_bank_bf16 from prepare_token_bank; temp float or 0-dim
    parameter (learnable in the reference: self.temp)."""
    if fusion_feats.device.type != "cuda":
        raise RuntimeError("spn4cir_amd runs on an MI355X (device='cuda'); there is no CPU path")
    labels = target_indexs.to(device=fusion_feats.device, dtype=torch.int64)
    return {"loss_qtc": _TokMaxLoss.apply(fusion_feats, temp, target_bank_bf16, labels)}
