"""Oracle: AdamW step as the reference configures it.

clip4cir/train_negplus.py:77-83: torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-7) with
the torch default weight_decay=0.01, amsgrad=False.  torch.optim.AdamW (third party) is
restated from its published update rule (decoupled weight decay):

    p   <- p * (1 - lr*wd)
    m   <- b1*m + (1-b1)*g ;  v <- b2*v + (1-b2)*g*g
    p   <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
"""
import math
import torch


def adamw_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-7, wd=0.01):
    """In-place on p, m, v (fp32 tensors); ``step`` is the 1-based step count."""
    p.mul_(1.0 - lr * wd)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p
