"""Host side of the BLIP fusion encoder (blip4cir/med.py BertModel mode='multimodal' + text_proj,
blip_cir.py:82-98) and the BLIP flavour of the stage-2 model (blip4cir/models.py:16-121).

Parameters live in one flat fp32 buffer (layout spn_fusion_layout); the reference's separate
query/key/value Linears are packed row-wise into one GEMM weight, and `named_views()` exposes every
tensor under its BertModel state-dict key (views into the packed buffers)."""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check, lib
from .ops import _p, _stream


def fusion_cfg_from_state_dict(sd, prefix=""):
    W = sd[prefix + "embeddings.word_embeddings.weight"].shape[1]
    layers = len({k[len(prefix):].split(".")[2] for k in sd if k.startswith(prefix + "encoder.layer.")})
    return dict(hidden=W, heads=W // 64, layers=layers, vocab=sd[prefix + "embeddings.word_embeddings.weight"].shape[0],
                max_pos=sd[prefix + "embeddings.position_embeddings.weight"].shape[0],
                intermediate=sd[prefix + "encoder.layer.0.intermediate.dense.weight"].shape[0],
                enc_width=sd[prefix + "encoder.layer.0.crossattention.self.key.weight"].shape[1])


class FusionEncoder:
    def __init__(self, hidden, layers, heads, intermediate, enc_width, proj_dim, vocab, max_pos, device="cuda"):
        if heads * 64 != hidden:
            raise ValueError("head_dim must be 64 (med_config.json: 768 / 12)")
        self.W, self.layers, self.H, self.I, self.E = hidden, layers, heads, intermediate, enc_width
        self.Dp, self.vocab, self.max_pos = proj_dim, vocab, max_pos
        self.device = torch.device(device)
        self._lay = _lib.FusionLayout()
        check(lib().spn_fusion_layout(C.byref(self._cfg(1, 1, 1)), C.byref(self._lay)), "fusion_layout")
        self.n_params = int(self._lay.n_params)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.wbf16 = torch.zeros(int(self._lay.n_bf16), dtype=torch.bfloat16, device=self.device)
        self._acts = self._ws = self._key = self._last = None
        self._stale = True

    def _cfg(self, B, L, S):
        return _lib.FusionCfg(B, L, S, self.W, self.H, self.layers, self.I, self.E, self.Dp, self.vocab, self.max_pos)

    def spans(self):
        """[(BertModel key (+ text_proj.*), offset, shape)]; q/k/v rows of the packed weights are separate views."""
        lay, W, I, E = self._lay, self.W, self.I, self.E
        lo = list(lay.layer_off)
        out = [("embeddings.word_embeddings.weight", lay.word, (self.vocab, W)),
               ("embeddings.position_embeddings.weight", lay.pos, (self.max_pos, W)),
               ("embeddings.LayerNorm.weight", lay.emb_ln_g, (W,)), ("embeddings.LayerNorm.bias", lay.emb_ln_b, (W,))]
        for l in range(self.layers):
            b = lay.layers + lay.layer_size * l
            p = f"encoder.layer.{l}."
            for i, n in enumerate(("query", "key", "value")):
                out.append((p + f"attention.self.{n}.weight", b + lo[0] + i * W * W, (W, W)))
                out.append((p + f"attention.self.{n}.bias", b + lo[1] + i * W, (W,)))
            out += [(p + "attention.output.dense.weight", b + lo[2], (W, W)), (p + "attention.output.dense.bias", b + lo[3], (W,)),
                    (p + "attention.output.LayerNorm.weight", b + lo[4], (W,)),
                    (p + "attention.output.LayerNorm.bias", b + lo[5], (W,)),
                    (p + "crossattention.self.query.weight", b + lo[6], (W, W)),
                    (p + "crossattention.self.query.bias", b + lo[7], (W,))]
            for i, n in enumerate(("key", "value")):
                out.append((p + f"crossattention.self.{n}.weight", b + lo[8] + i * W * E, (W, E)))
                out.append((p + f"crossattention.self.{n}.bias", b + lo[9] + i * W, (W,)))
            out += [(p + "crossattention.output.dense.weight", b + lo[10], (W, W)),
                    (p + "crossattention.output.dense.bias", b + lo[11], (W,)),
                    (p + "crossattention.output.LayerNorm.weight", b + lo[12], (W,)),
                    (p + "crossattention.output.LayerNorm.bias", b + lo[13], (W,)),
                    (p + "intermediate.dense.weight", b + lo[14], (I, W)), (p + "intermediate.dense.bias", b + lo[15], (I,)),
                    (p + "output.dense.weight", b + lo[16], (W, I)), (p + "output.dense.bias", b + lo[17], (W,)),
                    (p + "output.LayerNorm.weight", b + lo[18], (W,)), (p + "output.LayerNorm.bias", b + lo[19], (W,))]
        out += [("text_proj.weight", lay.proj_w, (self.Dp, W)), ("text_proj.bias", lay.proj_b, (self.Dp,))]
        return [(k, int(o), s) for k, o, s in out]

    def named_views(self, flat=None):
        flat = self.params if flat is None else flat
        views = {}
        for key, off, shape in self.spans():
            n = 1
            for s in shape:
                n *= s
            views[key] = flat[off:off + n].view(shape)
        return views

    def load_state_dict(self, sd, prefix=""):
        with torch.no_grad():
            for key, v in self.named_views().items():
                v.copy_(sd[prefix + key].to(self.device, torch.float32))
        self._stale = True

    def mark_stale(self):
        self._stale = True

    def forward(self, ids, mask, enc):
        """ids int32 [B,L], mask int32 [B,L] or None, enc fp32 [B,S,E] (device) -> text_proj output fp32 [B,Dp]."""
        B, L = ids.shape
        S = enc.shape[1]
        cfg = self._cfg(B, L, S)
        if self._stale:
            check(lib().spn_fusion_refresh_bf16(C.byref(cfg), _p(self.params), _p(self.wbf16), _stream()), "fusion_refresh")
            self._stale = False
        if self._key != (B, L, S):
            self._acts = torch.empty(lib().spn_fusion_act_bytes(C.byref(cfg)), dtype=torch.uint8, device=self.device)
            self._ws = torch.empty(lib().spn_fusion_ws_bytes(C.byref(cfg)), dtype=torch.uint8, device=self.device)
            self._key = (B, L, S)
        ids = ids.to(self.device, torch.int32).contiguous()
        mask = None if mask is None else mask.to(self.device, torch.int32).contiguous()
        enc = enc.to(self.device, torch.float32).contiguous()
        out = torch.empty(B, self.Dp, dtype=torch.float32, device=self.device)
        check(lib().spn_fusion_fwd(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(mask), _p(enc), _p(self._acts),
                                   _p(out), _stream()), "fusion_fwd")
        self._last = (ids, cfg)
        return out

    def backward(self, dproj):
        ids, cfg = self._last
        dproj = dproj.contiguous()
        check(lib().spn_fusion_bwd(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(self._acts), _p(dproj),
                                   _p(self.grads), _p(self._ws), self._ws.numel(), _stream()), "fusion_bwd")
        return self.grads


class BlipBankStep:
    """blip4cir/models.py:95-121 on the kernels: q = normalize(text_proj(fusion(ref_tokens, text)[:,0])),
    loss = CE(q @ target_bank.T / tau).  tau is learnable there (nn.Parameter, models.py:29): its gradient
    is dL/dtau = -(1/tau) * sum_b <q_b, dL/dq_b> and is returned next to the flat parameter gradient."""

    def __init__(self, encoder, tau=0.03, label_smoothing=0.0):
        self.enc, self.tau, self.ls = encoder, tau, label_smoothing

    def step(self, ids, mask, ref_tokens, bank_bf16, labels, grad_scale=None):
        proj = self.enc.forward(ids, mask, ref_tokens)
        q, qb, inv = ops.combine_l2norm_fwd(None, None, proj)
        M = bank_bf16.shape[0]
        stats = ops.bank_stats_fwd(qb, bank_bf16, labels, 1.0 / self.tau)
        lse, row, mean = ops.bank_loss_finalize(stats, M, self.ls)
        B = ids.shape[0]
        dq = ops.bank_grad_q(qb, bank_bf16, labels, 1.0 / self.tau, lse, (grad_scale or 1.0) / B, M_total=M,
                             label_smoothing=self.ls)[:, :self.enc.Dp].contiguous()
        dtau = -(q * dq).sum() / self.tau
        dproj = ops.combine_l2norm_bwd(q, inv, dq)
        grads = self.enc.backward(dproj)
        return mean, grads, dtau, q
