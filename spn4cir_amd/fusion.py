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

    def _cfg(self, B, L, S, T=0):
        return _lib.FusionCfg(B, L, S, self.W, self.H, self.layers, self.I, self.E, self.Dp, self.vocab, self.max_pos, T)

    @staticmethod
    def _prefix_lengths(mask):
        """Caption lengths of a HOST attention mask [B, L] if every row is ones-then-zeros with at least one 1 (the tokenizer's
        right padding, blip.py:189-194), else None.  A device mask is never inspected (that would synchronise the step)."""
        if mask is None or mask.is_cuda:
            return None
        m = mask.to(torch.int64)
        lens = m.sum(1)
        if int(lens.min()) < 1 or not bool((m == (torch.arange(m.shape[1])[None, :] < lens[:, None])).all()):
            return None
        return lens

    def _stage_cu(self, cu):
        """cu int32 [B + 1] on the host -> the same values in a pinned staging buffer that is safe to upload asynchronously."""
        if not torch.cuda.is_available():
            return cu
        ring = self.__dict__.setdefault("_cu_ring", [])
        n = cu.numel()
        if not ring or ring[0][0].numel() != n:
            ring.clear()
            ring.extend([torch.empty(n, dtype=torch.int32).pin_memory(), None] for _ in range(8))
            self._cu_next = 0
        slot = ring[self._cu_next]
        self._cu_next = (self._cu_next + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()                      # the upload that last used this buffer has run (eight steps ago)
        slot[0].copy_(cu)
        self._cu_pending = slot
        return slot[0]

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

    def is_stale(self):
        """True when the bf16 GEMM operands no longer match the fp32 masters: flagged explicitly (mark_stale) or the
        flat parameter buffer was written in place through any view since the last refresh - torch bumps the shared
        version counter for that, which is how an external `optimizer.step()` on the exposed nn.Parameters
        (train_negplus.py:121-123) is noticed without a parameters_changed() call."""
        return self._stale or self.params._version != getattr(self, "_seen_version", -1)

    def forward(self, ids, mask, enc=None, token_bank=None, token_idx=None, pack=None):
        """ids int32 [B,L], mask int32 [B,L] or None -> text_proj output fp32 [B,Dp].
        pack (default: on whenever possible): with a HOST mask of right-padded captions (what the tokenizer returns) only the
        unmasked text rows are materialised (spn_fusion_cfg.T; same features and gradients - a padded position's key is masked in
        every self-attention); a device mask, a shape spn_fusion_packed_ok rejects, or pack=False run the dense B x L rows.
        The reference image tokens are either
        `enc` fp32 [B,S,E] (device; cast to bf16 inside) or - the resident-bank form of the training step - rows `token_idx`
        (int64 [B], device) of `token_bank` bf16 [N,S,E] on the device, gathered by the library straight into the K/V
        projections' operand (spn_fusion_fwd_bank; blip4cir/models.py:97-100)."""
        B, L = ids.shape
        if (enc is None) == (token_bank is None):
            raise ValueError("give the image tokens either as enc= or as token_bank= + token_idx=")
        if token_bank is not None:
            if token_bank.dtype != torch.bfloat16 or not token_bank.is_cuda or token_bank.dim() != 3 or not token_bank.is_contiguous():
                raise ValueError("token_bank must be a contiguous bf16 [N, S, E] device tensor (ops.token_bank_bf16)")
            if token_idx is None or token_idx.dtype != torch.int64 or not token_idx.is_cuda or token_idx.numel() != B:
                raise ValueError("token_idx must be an int64 [B] device tensor")
            if token_bank.shape[2] != self.E:
                raise ValueError(f"token_bank width {token_bank.shape[2]} != encoder width {self.E}")
            S = token_bank.shape[1]
        else:
            S = enc.shape[1]
        cfg0 = self._cfg(B, L, S)
        lens = None
        if pack is not False and L <= 128 and lib().spn_fusion_packed_ok(C.byref(cfg0)):
            lens = self._prefix_lengths(mask)
        if pack and lens is None:
            raise ValueError("pack=True needs a host (CPU) attention mask of right-padded captions and a shape spn_fusion_packed_ok accepts")
        if lens is not None:
            cu = torch.zeros(B + 1, dtype=torch.int32)
            cu[1:] = lens.cumsum(0)
            cfg = self._cfg(B, L, S, int(cu[-1]))
            # the mask slot carries cu_seqlens (spn4cir_hip.h).  They go up through a small ring of pinned staging buffers, each
            # guarded by an event recorded behind its upload: a host that enqueues several steps ahead never overwrites prefix sums
            # a queued copy has yet to read, and no step pays for a pinned allocation (round 5 took a fresh `pin_memory()` tensor
            # per step: usually recycled by torch's caching host allocator, but a slow path of several milliseconds when it was not
            # - config 4's packed step measured 8.4 ms or 13 ms from run to run)
            mask = self._stage_cu(cu)
        else:
            cfg = cfg0
        if self.is_stale():
            check(lib().spn_fusion_refresh_bf16(C.byref(cfg), _p(self.params), _p(self.wbf16), _stream()), "fusion_refresh")
            self._stale = False
            self._seen_version = self.params._version
        if self._key != (B, L, S):
            # one arena per shape, sized for the worst batch of that shape: the dense rows AND the fullest packed batch (T = B * L -
            # all captions of equal length, or B = 1 - whose index arrays come on top of the dense rows)
            full = self._cfg(B, L, S, B * L)
            self._acts = ops.scratch_bytes(max(lib().spn_fusion_act_bytes(C.byref(c)) for c in (cfg0, full)), self.device)
            self._ws = ops.scratch_bytes(max(lib().spn_fusion_ws_bytes(C.byref(c)) for c in (cfg0, full)), self.device)
            self._key = (B, L, S)
        if lib().spn_fusion_act_bytes(C.byref(cfg)) > self._acts.numel() or lib().spn_fusion_ws_bytes(C.byref(cfg)) > self._ws.numel():
            raise RuntimeError("fusion arena smaller than this batch needs")          # cannot happen; never write past the arena
        ids = ids.to(self.device, torch.int32).contiguous()
        mask = None if mask is None else mask.to(self.device, torch.int32, non_blocking=True).contiguous()
        pend = self.__dict__.pop("_cu_pending", None)
        if pend is not None:                               # the asynchronous upload of the staged prefix sums is enqueued: fence it
            pend[1] = torch.cuda.Event()
            pend[1].record()
        out = torch.empty(B, self.Dp, dtype=torch.float32, device=self.device)
        if token_bank is not None:
            check(lib().spn_fusion_fwd_bank(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(mask), _p(token_bank),
                                            token_bank.shape[0], _p(token_idx.contiguous()), _p(self._acts), _p(out), _stream()),
                  "fusion_fwd_bank")
        else:
            enc = enc.to(self.device, torch.float32).contiguous()
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


    def layer_spans(self):
        """Flat-gradient ranges in the order backward_phased finishes them: [head (text_proj), layer L-1, ..., layer 0,
        tail (embeddings)]."""
        lay = self._lay
        out = [(int(lay.proj_w), self.n_params)]
        for l in reversed(range(self.layers)):
            b = int(lay.layers + lay.layer_size * l)
            out.append((b, b + int(lay.layer_size)))
        out.append((0, int(lay.layers)))
        return out

    def backward_phased(self, dproj, on_span_ready, groups=None):
        """backward() in phases (spn_fusion_bwd_phase): on_span_ready(start, end) is called right after the launches that
        finish the flat-gradient range [start, end) have been enqueued - head, then the layers in `groups` (block counts from
        the top, summing to `layers`; a group's weight gradients are one set of grouped launches behind its data path), then
        the embeddings: the DDP bucket hook, as TextTower.backward_phased."""
        ids, cfg = self._last
        dproj = dproj.contiguous()
        groups = groups or [self.layers]
        if sum(groups) != self.layers or any(g <= 0 for g in groups):
            raise ValueError(f"groups {groups} must split {self.layers} layers")
        spans = self.layer_spans()

        def phase(k, lo, hi):
            check(lib().spn_fusion_bwd_phase(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(self._acts), _p(dproj),
                                             _p(self.grads), _p(self._ws), self._ws.numel(), k, lo, hi, _stream()),
                  "fusion_bwd_phase")
        phase(0, 0, 0)
        on_span_ready(*spans[0])
        hi = self.layers
        for g in groups:
            lo = hi - g
            phase(1, lo, hi)
            for l in reversed(range(lo, hi)):
                on_span_ready(*spans[1 + (self.layers - 1 - l)])
            hi = lo
        phase(2, 0, 0)
        on_span_ready(*spans[-1])
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
        saved = ops.bank_logits_buffer(qb.shape[0], M, qb.device)
        stats = ops.bank_stats_fwd(qb, bank_bf16, labels, 1.0 / self.tau, save=saved)
        lse, row, mean = ops.bank_loss_finalize(stats, M, self.ls)
        B = ids.shape[0]
        dq = ops.bank_grad_q(qb, bank_bf16, labels, 1.0 / self.tau, lse, (grad_scale or 1.0) / B, M_total=M,
                             label_smoothing=self.ls, saved=saved)[:, :self.enc.Dp].contiguous()
        dtau = -(q * dq).sum() / self.tau
        dproj = ops.combine_l2norm_bwd(q, inv, dq)
        grads = self.enc.backward(dproj)
        return mean, grads, dtau, q


class BlipStage2Trainer:
    """blip4cir stage-2 loop (blip4cir/train.py:110-129) on one GPU or data-parallel over RCCL (BASELINE config 4):
    triplets sharded across ranks, fusion encoder replicated, bank replicated or sharded exactly as for the CLIP
    path (spn4cir_amd.distributed.BankLossDP), flat-gradient all-reduce, fused AdamW on the flat buffer and a
    scalar AdamW for the learnable temperature (models.py:29).  The reference optimises [encoder, tau] with
    AdamW(lr, betas (0.9, 0.999), eps 1e-7) and default weight decay."""

    def __init__(self, encoder, tau=0.03, lr=5e-6, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01, group=None,
                 bank_mode="auto", label_smoothing=0.0, learn_tau=True, bucket_elems=8 << 20, optim="replicated"):
        from . import distributed as dp
        self.enc, self.group = encoder, group
        self._bank_mode_arg = bank_mode      # "auto": replicated below 10^6 bank rows (set_bank), as bench.py --bank-mode auto
        self.lr, self.betas, self.eps, self.wd, self.ls = lr, betas, eps, weight_decay, label_smoothing
        self.world, self.rank = dp._world(group)
        first = "replicated" if bank_mode == "auto" else bank_mode
        self.loss_dp = dp.BankLossDP(ops, group, first if (self.world > 1 or dp._FORCE) else "replicated")
        self._shard_range = dp.shard_range
        self.m = torch.zeros_like(encoder.params)
        self.v = torch.zeros_like(encoder.params)
        # optim="sharded": the sharded optimizer step of trainer.Stage2Trainer (ZeRO-1 shape) - a bucket's reduced gradient chunk is
        # updated by its owner, the masters are all-gathered; the temperature (one scalar) stays replicated
        if optim not in ("replicated", "sharded"):
            raise ValueError(optim)
        self.optim = optim if (self.world > 1 or dp._FORCE) else "replicated"
        if self.optim == "sharded":
            self.reducer = dp.GradBucketReducer(encoder.grads, group, bucket_elems, shard_update=self._update_range,
                                                flat_params=encoder.params)
        else:
            self.reducer = dp.GradBucketReducer(encoder.grads, group, bucket_elems)
        dev = encoder.device
        self.tau = torch.tensor([float(tau)], dtype=torch.float32, device=dev)      # learnable temperature
        self.learn_tau = learn_tau
        self._tau_m = torch.zeros(1, dtype=torch.float32, device=dev)
        self._tau_v = torch.zeros(1, dtype=torch.float32, device=dev)
        self._tau_g = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0
        self._bank, self._m_begin, self._M_total = None, 0, 0

    def _update_range(self, lo, hi, grad=None):
        e = self.enc
        ops.adamw_step(e.params[lo:hi], e.grads[lo:hi] if grad is None else grad, self.m[lo:hi], self.v[lo:hi], self.step_count,
                       self.lr, self.betas, self.eps, self.wd)

    def set_bank(self, target_bank, bank_dtype="bf16"):
        """target_bank fp32 [M, Dp], L2-normalised rows; sharded mode keeps only this rank's rows on the device."""
        dev = self.enc.device
        self._M_total = target_bank.shape[0]
        if self._bank_mode_arg == "auto" and self.world > 1:
            # a 30 000 x 256 bank is 15 MB: sharding it puts three latency-bound collectives on the critical path to save a
            # few microseconds of bank kernel; the all-gather variant pays from ~10^6 rows
            self.loss_dp.mode = "sharded" if self._M_total >= 1000000 else "replicated"
        if self.loss_dp.mode == "sharded" and self.world > 1:
            b, e = self._shard_range(self._M_total, self.world, self.rank)
            self._m_begin = b
            self._bank = ops.prepare_bank(target_bank[b:e].to(dev, torch.float32).contiguous(), bank_dtype)
        else:
            self._m_begin = 0
            self._bank = ops.prepare_bank(target_bank.to(dev, torch.float32), bank_dtype)

    def set_token_bank(self, refer_bank):
        """The reference-token bank of blip4cir/models.py:45-89 ([N, 577, W] fp32, host RAM there) as a bf16 image on the device
        (26.6 GB at 30 000 x 577 x 768): step(..., token_idx=) then gathers the batch's rows on the device."""
        self._token_bank = ops.token_bank_bf16(refer_bank, self.enc.device)

    def step(self, ids, mask, ref_tokens, labels, token_idx=None):
        """ids/mask int32 [B_local, L], labels int64 [B_local] global bank rows; the reference image tokens either as
        ref_tokens fp32 [B_local, S, E] or (ref_tokens=None) as rows token_idx int64 [B_local] of the bank given to
        set_token_bank.  Returns the global mean loss (1-element device tensor)."""
        enc = self.enc
        from .trainer import wgrad_groups
        if ref_tokens is None:
            if token_idx is None or getattr(self, "_token_bank", None) is None:
                raise ValueError("step(ref_tokens=None) needs set_token_bank(...) and token_idx=")
            proj = enc.forward(ids, mask, token_bank=self._token_bank, token_idx=token_idx.to(enc.device, torch.int64))
        else:
            proj = enc.forward(ids, mask, ref_tokens)
        q, _, inv = ops.combine_l2norm_fwd(None, None, proj)
        # the temperature is a parameter that lives on the device (models.py:29): logits = (q / tau) . bank with the bank
        # kernels at inv_tau = 1, every factor of tau applied by device-side scalars - no host read, so the host keeps
        # enqueueing the next step while this one runs
        qs = ops.scale_cast_bf16(q, self.tau, reciprocal=True, ldo=self._bank.shape[1])
        ctx = self.loss_dp.forward(qs, labels, self._bank, self._m_begin, self._M_total, 1.0, self.ls)
        dqk = self.loss_dp.backward(ctx)                                  # d loss / d (q / tau), [B, ldq]
        inv_tau = ops.tau_grad(q, dqk, self.tau, self._tau_g if self.learn_tau else None)     # d loss / d tau and 1 / tau, one launch
        dproj = ops.combine_l2norm_bwd(q, inv, dqk if dqk.shape[1] == enc.Dp else dqk[:, :enc.Dp].contiguous(), scale=inv_tau)
        groups = wgrad_groups(enc.layers, self.world)
        self.step_count += 1
        if self.optim == "sharded":
            enc.backward_phased(dproj, self.reducer.on_span_ready, groups if self.world > 1 else None)
            inflight = self.reducer.finish_unsharded()
            for lo, hi in self.reducer.complement_spans(enc.params.numel()):      # buckets the direct exchange could not split
                self._update_range(lo, hi)
            for w in inflight:
                w.wait()
        else:
            enc.backward_phased(dproj, self.reducer.on_span_ready, groups if self.world > 1 else None)
            self.reducer.finish()
            ops.adamw_step(enc.params, enc.grads, self.m, self.v, self.step_count, self.lr, self.betas, self.eps, self.wd)
        enc.mark_stale()
        if self.learn_tau:
            if self.world > 1:
                torch.distributed.all_reduce(self._tau_g, group=self.group)
            ops.adamw_step(self.tau, self._tau_g, self._tau_m, self._tau_v, self.step_count, self.lr, self.betas, self.eps,
                           self.wd)
        return ctx["loss"]
