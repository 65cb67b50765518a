"""Drop-in for `torch.optim.AdamW` over the parameters a spn4cir_amd model exposes (train_negplus.py:77-84 builds
`optim.AdamW([{'params': ..., 'lr': ..., 'betas': (0.9, 0.999), 'eps': 1e-7}])`; change that one line to
`spn4cir_amd.optim.AdamW(...)` with the same arguments).

The models' nn.Parameters are slices of ONE flat fp32 buffer per tower and their `.grad`s are slices of one flat gradient
buffer (gradsink.py), so the whole update is a single launch of the fused AdamW kernel (spn_adamw_step_scaled, the kernel
Stage2Trainer uses) instead of torch's ~10 multi-tensor passes (2.9 ms -> 0.6 ms per step for ViT-L/14's text tower).
Same arithmetic as torch.optim.AdamW (decoupled weight decay, bias correction, eps outside the square root).  The step
count that enters the bias correction lives on the DEVICE (one fp32 counter shared by every parameter's state["step"],
advanced by spn_adamw_tick only when found_inf == 0): a step GradScaler skips on overflow does not advance it, exactly as
torch - which does not call optimizer.step() on such a step - leaves its per-parameter counters alone.

torch.amp.GradScaler: the class declares `_step_supports_amp_scaling`, so `scaler.step(optimizer)` hands over
`optimizer.grad_scale` / `optimizer.found_inf` (device tensors) and the kernel unscales the gradients and skips the step on
overflow itself - no `unscale_` pass over the gradients and no host synchronisation.

Parameters that are not part of such a flat run (or whose gradients are not the matching slices) are updated one launch
per tensor with the same kernel, so the class works for any fp32 CUDA parameter list."""
import ctypes as C

import torch

from ._lib import check, lib
from .ops import _p, _stream


def _flat_view(t, numel):
    """1-D fp32 tensor over t's storage starting at t's first element."""
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), (numel,), (1,))


class AdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, on_step=None):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._on_step = on_step             # e.g. model.parameters_changed (the models also notice the update themselves)
        self._runs = None                   # per group: [dict(params, p_flat, m, v, numel)]
        self._step_dev = None               # 0-dim fp32 device counter of APPLIED steps (shared by every state["step"])

    # -------------------------------------------------------------------------------- layout
    def _build_runs(self):
        self._runs = []
        dev = next(p.device for g in self.param_groups for p in g["params"])
        self._step_dev = torch.zeros((), dtype=torch.float32, device=dev)
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            for p in ps:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise TypeError("spn4cir_amd.optim.AdamW: parameters must be contiguous fp32 CUDA tensors")
            ps.sort(key=lambda p: (p.untyped_storage().data_ptr(), p.storage_offset()))
            runs, cur = [], []
            for p in ps:
                if cur and (p.untyped_storage().data_ptr() == cur[-1].untyped_storage().data_ptr()
                            and p.storage_offset() == cur[-1].storage_offset() + cur[-1].numel()):
                    cur.append(p)
                else:
                    if cur:
                        runs.append(cur)
                    cur = [p]
            if cur:
                runs.append(cur)
            out = []
            for r in runs:
                n = sum(p.numel() for p in r)
                m = torch.zeros(n, dtype=torch.float32, device=r[0].device)
                v = torch.zeros(n, dtype=torch.float32, device=r[0].device)
                off = 0
                for p in r:                                  # torch-compatible per-parameter state (views of the run's state)
                    st = self.state[p]
                    st["step"] = self._step_dev
                    st["exp_avg"] = m[off:off + p.numel()].view_as(p)
                    st["exp_avg_sq"] = v[off:off + p.numel()].view_as(p)
                    off += p.numel()
                out.append(dict(params=r, p_flat=_flat_view(r[0], n), m=m, v=v, numel=n))
            self._runs.append(out)

    def load_state_dict(self, state_dict):
        """torch semantics; the loaded per-parameter moments are copied into the flat state of the runs."""
        super().load_state_dict(state_dict)
        loaded = {p: dict(st) for p, st in self.state.items()}
        self._build_runs()                                   # fresh flat state, self.state[p] re-pointed at its views
        steps = 0.0
        for p, old in loaded.items():
            st = self.state[p]
            if "exp_avg" in old:
                st["exp_avg"].copy_(old["exp_avg"].to(st["exp_avg"].device, torch.float32).view_as(st["exp_avg"]))
                st["exp_avg_sq"].copy_(old["exp_avg_sq"].to(st["exp_avg_sq"].device, torch.float32).view_as(st["exp_avg_sq"]))
            if "step" in old:
                steps = max(steps, float(old["step"]))
        self._step_dev.fill_(steps)

    @staticmethod
    def _grads_flat(run):
        """The run's gradients as one flat tensor if they are consecutive slices of one buffer, else None."""
        g0 = run["params"][0].grad
        if g0 is None or g0.dtype != torch.float32 or not g0.is_contiguous():
            return None
        base, off = g0.untyped_storage().data_ptr(), g0.storage_offset()
        for p in run["params"]:
            g = p.grad
            if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device
                    or g.untyped_storage().data_ptr() != base or g.storage_offset() != off):
                return None
            off += p.numel()
        return _flat_view(g0, run["numel"])

    # -------------------------------------------------------------------------------- step
    def _launch(self, p, g, m, v, group, scale, found):
        b1, b2 = group["betas"]
        check(lib().spn_adamw_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), float(group["lr"]), float(b1), float(b2),
                                       float(group["eps"]), float(group["weight_decay"]), _p(self._step_dev), _p(scale),
                                       _p(found), _stream()), "adamw_step_dev")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._runs is None:
            self._build_runs()
        scale = getattr(self, "grad_scale", None)          # set by torch.amp.GradScaler.step around this call
        found = getattr(self, "found_inf", None)
        for t in (scale, found):
            if t is not None and (t.dtype != torch.float32 or not t.is_cuda or t.numel() != 1):
                raise TypeError("grad_scale / found_inf must be 1-element fp32 CUDA tensors")
        check(lib().spn_adamw_tick(_p(self._step_dev), _p(found), _stream()), "adamw_tick")
        for group, runs in zip(self.param_groups, self._runs):
            for run in runs:
                if all(p.grad is None for p in run["params"]):
                    continue
                g = self._grads_flat(run)
                if g is not None:
                    self._launch(run["p_flat"], g, run["m"], run["v"], group, scale, found)
                else:                                        # gradients live elsewhere: one launch per tensor
                    off = 0
                    for p in run["params"]:
                        n = p.numel()
                        if p.grad is not None:
                            gg = p.grad.contiguous().float()
                            self._launch(_flat_view(p, n), gg.view(-1), run["m"][off:off + n], run["v"][off:off + n], group,
                                         scale, found)
                        off += n
                run["params"][0].view(-1)[:0].zero_()       # bump the buffer's version counter: the models re-derive
                                                             # their bf16 operands when it has moved
        if self._on_step is not None:
            self._on_step()
        return loss
