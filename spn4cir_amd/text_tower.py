"""Host side of the CLIP text tower (CLIP.encode_text, clip4cir/clip/model.py:345-358).

All parameters live in ONE flat fp32 device buffer (layout: spn_text_layout in
include/spn4cir_hip.h) so that the optimizer is a single fused launch and the DDP gradient
all-reduce runs over contiguous buckets; `named_views()` exposes it under the reference's
state-dict keys.  bf16 mirrors (+ transposes) of the GEMM weights are refreshed after every
optimizer step.  Forward/backward are single C-ABI calls that enqueue the whole launch chain.
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import check, lib
from ._lib import env as _lib_env
from .ops import _p, _stream

_BLOCK_KEYS = ["ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
               "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
               "mlp.c_proj.weight", "mlp.c_proj.bias"]


def text_cfg_from_state_dict(sd, prefix=""):
    """Shape inference as build_model does it (clip4cir/clip/model.py:420-426)."""
    width = sd[prefix + "ln_final.weight"].shape[0]
    layers = len({k[len(prefix):].split(".")[2] for k in sd if k.startswith(prefix + "transformer.resblocks.")})
    return dict(width=width, layers=layers, heads=width // 64, embed_dim=sd[prefix + "text_projection"].shape[1],
                vocab=sd[prefix + "token_embedding.weight"].shape[0], ctx=sd[prefix + "positional_embedding"].shape[0])


class TextTower:
    def __init__(self, width, layers, heads, embed_dim, vocab=49408, ctx=77, device="cuda"):
        if heads * 64 != width:
            raise ValueError("CLIP text towers use head_dim 64 (heads = width // 64, clip/model.py:425)")
        self.width, self.layers, self.heads = width, layers, heads
        self.embed_dim, self.vocab, self.ctx = embed_dim, vocab, ctx
        self.device = torch.device(device)
        self._lay = _lib.TextLayout()
        check(lib().spn_text_layout(C.byref(self._cfg(1, ctx)), C.byref(self._lay)), "text_layout")
        self.n_params = int(self._lay.n_params)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.wbf16 = torch.zeros(int(self._lay.n_bf16), dtype=torch.bfloat16, device=self.device)
        self._acts = None
        self._acts_key = None
        self._ws = None
        self._last = None
        self._last_T = 0
        self._stale = True

    # ------------------------------------------------------------------ layout
    def _cfg(self, B, L, T=0):
        # pool: the last block's row-wise half on the pooled (EOT) rows only (spn_text_cfg.pool) - on for the feature path
        # (forward / backward*), off for forward_tokens, whose ln_final reads every row; a backward uses its forward's setting
        return _lib.TextCfg(B, L, self.ctx, self.width, self.heads, self.layers, self.embed_dim, self.vocab, T,
                            getattr(self, "_pool", 1))

    def spans(self):
        """[(clip state-dict key, offset, shape)] in flat-buffer order."""
        lay, W, D = self._lay, self.width, self.embed_dim
        out = [("token_embedding.weight", lay.tok, (self.vocab, W)), ("positional_embedding", lay.pos, (self.ctx, W))]
        shapes = [(W,), (W,), (3 * W, W), (3 * W,), (W, W), (W,), (W,), (W,), (4 * W, W), (4 * W,), (W, 4 * W), (W,)]
        for l in range(self.layers):
            base = lay.blocks + lay.block_size * l
            for j, key in enumerate(_BLOCK_KEYS):
                out.append((f"transformer.resblocks.{l}.{key}", base + lay.block_off[j], shapes[j]))
        out += [("ln_final.weight", lay.lnf_g, (W,)), ("ln_final.bias", lay.lnf_b, (W,)),
                ("text_projection", lay.text_proj, (W, D))]
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

    def layer_spans(self):
        """Flat (start, end) ranges in backward-completion order (last layer first): DDP buckets."""
        lay = self._lay
        tail = (int(lay.lnf_g), self.n_params)
        blocks = [(int(lay.blocks + lay.block_size * l), int(lay.blocks + lay.block_size * (l + 1)))
                  for l in reversed(range(self.layers))]
        head = (0, int(lay.blocks))
        return [tail] + blocks + [head]

    def load_clip_state_dict(self, sd, prefix=""):
        views = self.named_views()
        with torch.no_grad():
            for key, v in views.items():
                v.copy_(sd[prefix + key].to(device=self.device, dtype=torch.float32))
        self._stale = True

    def refresh(self):
        """Rewrite the bf16 GEMM operands from the fp32 master weights."""
        check(lib().spn_text_refresh_bf16(C.byref(self._cfg(1, self.ctx)), _p(self.params), _p(self.wbf16), _stream()),
              "text_refresh_bf16")
        self._stale = False
        self._seen_version = self.params._version

    def mark_stale(self):
        self._stale = True

    def is_stale(self):
        """True when the bf16 GEMM operands no longer match the fp32 masters: flagged explicitly (mark_stale) or the
        flat parameter buffer was written in place through any view since the last refresh - torch bumps the shared
        version counter for that, which is how an external `optimizer.step()` on the exposed nn.Parameters
        (train_negplus.py:121-123) is noticed without a parameters_changed() call."""
        return self._stale or self.params._version != getattr(self, "_seen_version", -1)

    # ------------------------------------------------------------------ compute
    def _buffers(self, B, L, need_ws, T=0):
        """Activation arena and backward workspace, sized ONCE per batch capacity for the full context length (the dense
        [B, L_ctx] layout bounds every shorter or packed one): the packed callers cut the id matrix behind the longest
        caption, so L changes from batch to batch - keying the arenas on (B, L) dropped and re-allocated 6 + 3 GB at
        ViT-L/14, B = 256 whenever the length bucket moved.  They grow only when B does."""
        if self._acts_key is None or B > self._acts_key:
            self._acts = self._ws = None
            cap = self._cfg(B, self.ctx)
            self._acts = ops.scratch_bytes(lib().spn_text_act_bytes(C.byref(cap)), self.device)
            self._acts_key = B
        if need_ws and self._ws is None:
            cap = self._cfg(self._acts_key, self.ctx)
            self._ws = ops.scratch_bytes(lib().spn_text_ws_bytes(C.byref(cap)), self.device)
        return self._cfg(B, L, T)

    @staticmethod
    def cu_seqlens(ids_host):
        """Host helper for the packed mode: int32 [B+1] prefix sums of the live lengths (EOT position + 1,
        EOT = argmax of the ids as in clip/model.py:356) of a CPU id matrix, and their total."""
        ids_host = torch.as_tensor(ids_host)
        lens = ids_host.argmax(dim=-1).to(torch.int64) + 1
        cu = torch.zeros(ids_host.shape[0] + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
        return cu, int(cu[-1])

    @staticmethod
    def live_length(ids_host, multiple=16):
        """Host helper for the packed mode: the longest live caption of a CPU id matrix, rounded up to `multiple` and
        capped at its width.  Columns beyond it hold padding only (they are dead in the packed layout anyway), so the caller
        may pass `ids[:, :live_length]` - the towers take any L <= context length - and the whole-head attention kernels
        then run with ceil(L / 16) instead of ceil(77 / 16) waves per (caption, head)."""
        ids_host = torch.as_tensor(ids_host)
        longest = int(ids_host.argmax(dim=-1).max()) + 1
        return min(int(ids_host.shape[1]), (longest + multiple - 1) // multiple * multiple)

    def forward(self, ids, cu_seqlens=None, total_rows=0):
        """ids int32 [B, L] on device -> fp32 [B, D]; activations are kept for backward().

        cu_seqlens (device int32 [B+1]) + total_rows select the packed mode: only the rows up to each caption's
        EOT token are computed (the rest is dead under the causal mask), same features and gradients."""
        if ids.dtype != torch.int32 or not ids.is_cuda or not ids.is_contiguous():
            raise ValueError("ids must be a contiguous int32 device tensor")
        B, L = ids.shape
        if self.is_stale():
            self.refresh()
        T = int(total_rows) if cu_seqlens is not None else 0
        self._pool = 1
        cfg = self._buffers(B, L, False, T)
        feats = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        if cu_seqlens is not None:
            if cu_seqlens.dtype != torch.int32 or not cu_seqlens.is_cuda or cu_seqlens.numel() != B + 1 or T <= 0:
                raise ValueError("cu_seqlens must be a device int32 [B+1] tensor and total_rows its last entry")
            check(lib().spn_text_fwd_packed(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(cu_seqlens),
                                            _p(self._acts), _p(feats), _stream()), "text_fwd_packed")
        else:
            check(lib().spn_text_fwd(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(self._acts), _p(feats),
                                     _stream()), "text_fwd")
        self._last = ids
        self._last_T = T
        return feats

    def forward_exact(self, ids):
        """fp32-exact features (f32-input MFMA GEMMs, fp32 attention; inference only): for validation / retrieval,
        where the bf16 tower's ~1e-2 relative feature error can flip near-ties of a ranking."""
        if ids.dtype != torch.int32 or not ids.is_cuda or not ids.is_contiguous():
            raise ValueError("ids must be a contiguous int32 device tensor")
        B, L = ids.shape
        cfg = self._cfg(B, L)
        need = lib().spn_text_exact_ws_bytes(C.byref(cfg))
        ws = ops.scratch_bytes(need, self.device)
        feats = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        check(lib().spn_text_fwd_exact(C.byref(cfg), _p(self.params), _p(ids), _p(ws), ws.numel(), _p(feats), _stream()),
              "text_fwd_exact")
        return feats

    def backward(self, dfeats):
        """d(loss)/d(feats) fp32 [B, D] -> fills self.grads (overwrites) and returns it."""
        ids = self._last
        if ids is None:
            raise RuntimeError("backward() without a preceding forward()")
        B, L = ids.shape
        cfg = self._buffers(B, L, True, self._last_T)
        dfeats = dfeats.contiguous()
        check(lib().spn_text_bwd(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(self._acts), _p(dfeats),
                                 _p(self.grads), _p(self._ws), self._ws.numel(), _stream()), "text_bwd")
        return self.grads

    def forward_tokens(self, ids):
        """TG-CIR's text side (tgcir/models.py:127-137): -> (feats fp32 [B, D], tokens fp32 [B, L, W], tokens bf16):
        the pooled feature plus ln_final of EVERY position (padding rows included - dense layout only)."""
        if ids.dtype != torch.int32 or not ids.is_cuda or not ids.is_contiguous():
            raise ValueError("ids must be a contiguous int32 device tensor")
        B, L = ids.shape
        if self.is_stale():
            self.refresh()
        self._pool = 0
        cfg = self._buffers(B, L, False, 0)
        feats = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        tokens = torch.empty(B, L, self.width, dtype=torch.float32, device=self.device)
        tokens_b = torch.empty(B, L, self.width, dtype=torch.bfloat16, device=self.device)
        self._tok_stats = torch.empty(2, B * L, dtype=torch.float32, device=self.device)
        check(lib().spn_text_fwd_tokens(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(self._acts), _p(feats),
                                        _p(tokens), _p(tokens_b), _p(self._tok_stats[0]), _p(self._tok_stats[1]),
                                        _stream()), "text_fwd_tokens")
        self._last = ids
        self._last_T = 0
        return feats, tokens, tokens_b

    def backward_tokens(self, dfeats, dtokens):
        """Gradients of both outputs of forward_tokens -> fills self.grads (overwrites) and returns it."""
        ids = self._last
        if ids is None or getattr(self, "_tok_stats", None) is None:
            raise RuntimeError("backward_tokens() without a preceding forward_tokens()")
        B, L = ids.shape
        cfg = self._buffers(B, L, True, 0)
        dfeats, dtokens = dfeats.contiguous(), dtokens.contiguous()
        if dtokens.dtype != torch.float32 or dtokens.numel() != B * L * self.width:
            raise ValueError("dtokens must be fp32 [B, L, W]")
        check(lib().spn_text_bwd_tokens(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(ids), _p(self._acts), _p(dfeats),
                                        _p(dtokens), _p(self._tok_stats[0]), _p(self._tok_stats[1]), _p(self.grads),
                                        _p(self._ws), self._ws.numel(), _stream()), "text_bwd_tokens")
        return self.grads

    def backward_tokens_phased(self, dfeats, dtokens, on_span_ready, wgrad_groups=None):
        """backward_tokens() in phases (TG-CIR under data parallelism): on_span_ready(start, end) right after the launches that
        finish a flat-gradient range - head (ln_final + text_projection), then the blocks group by group (wgrad_groups: block
        counts from the top, each group's weight gradients in one grouped launch behind its data path), then the embeddings."""
        ids = self._last
        if ids is None or getattr(self, "_tok_stats", None) is None:
            raise RuntimeError("backward_tokens_phased() without a preceding forward_tokens()")
        B, L = ids.shape
        cfg = self._buffers(B, L, True, 0)
        dfeats, dtokens = dfeats.contiguous(), dtokens.contiguous()
        if dtokens.dtype != torch.float32 or dtokens.numel() != B * L * self.width:
            raise ValueError("dtokens must be fp32 [B, L, W]")
        ws, n = _p(self._ws), self._ws.numel()
        spans = self.layer_spans()
        check(lib().spn_text_bwd_tokens_head(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts), _p(dfeats), _p(dtokens),
                                             _p(self._tok_stats[0]), _p(self._tok_stats[1]), _p(self.grads), ws, n, _stream()),
              "text_bwd_tokens_head")
        on_span_ready(*spans[0])
        if (_lib_env("SPN_TN_GROUP", "1") or "1")[:1] == "0" or wgrad_groups is None:
            for i, l in enumerate(reversed(range(self.layers))):
                check(lib().spn_text_bwd_layer(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts), _p(self.grads), l,
                                               ws, n, _stream()), "text_bwd_layer")
                on_span_ready(*spans[1 + i])
        else:
            if sum(wgrad_groups) != self.layers or any(g <= 0 or g > 12 for g in wgrad_groups):
                raise ValueError(f"wgrad_groups {wgrad_groups} must split {self.layers} layers into groups of 1..12")
            hi = self.layers
            for g in wgrad_groups:
                lo = hi - g
                for l in reversed(range(lo, hi)):
                    check(lib().spn_text_bwd_layer_deferred(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts),
                                                            _p(self.grads), l, ws, n, _stream()), "text_bwd_layer_deferred")
                check(lib().spn_text_bwd_wgrad(C.byref(cfg), _p(self._acts), _p(self.grads), lo, hi, ws, n, _stream()),
                      "text_bwd_wgrad")
                for l in reversed(range(lo, hi)):
                    on_span_ready(*spans[1 + (self.layers - 1 - l)])
                hi = lo
        check(lib().spn_text_bwd_tail_tokens(C.byref(cfg), _p(ids), _p(self._acts), _p(self.grads), ws, n, _stream()),
              "text_bwd_tail_tokens")
        on_span_ready(*spans[-1])
        return self.grads

    def backward_phased(self, dfeats, on_span_ready, wgrad_groups=None, embed_early=False):
        """Same as backward(), but calls on_span_ready(start, end) right after the launches that
        finish the flat-gradient range [start, end) have been enqueued (tail+head params first, then
        each block from the last to the first, then the embeddings): the DDP bucket hook.

        wgrad_groups (list of block counts, last block first, summing to `layers`, each <= 12): the weight gradients of
        a group's blocks are deferred to ONE grouped launch behind the group's data path (spn_text_bwd_wgrad); the
        group's spans are reported after it.  None: every block computes its own (one grouped launch per block).
        embed_early (with wgrad_groups): the embedding gradients (phase 3; they only need the data path of block 0) are
        computed and reported BEFORE the last group's weight gradients, so that their all-reduce - 152 MB for ViT-L/14,
        a third of all gradient bytes - runs under that grouped launch instead of after the backward pass."""
        ids = self._last
        if ids is None:
            raise RuntimeError("backward_phased() without a preceding forward()")
        B, L = ids.shape
        cfg = self._buffers(B, L, True, self._last_T)
        dfeats = dfeats.contiguous()
        ws, n = _p(self._ws), self._ws.numel()
        spans = self.layer_spans()
        check(lib().spn_text_bwd_head(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts), _p(dfeats),
                                      _p(self.grads), ws, n, _stream()), "text_bwd_head")
        on_span_ready(*spans[0])
        if (_lib_env("SPN_TN_GROUP", "1") or "1")[:1] == "0":
            wgrad_groups = None                     # the library's workspace then holds no deferred buffers
        if wgrad_groups is None:
            for i, l in enumerate(reversed(range(self.layers))):
                check(lib().spn_text_bwd_layer(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts),
                                               _p(self.grads), l, ws, n, _stream()), "text_bwd_layer")
                on_span_ready(*spans[1 + i])
        else:
            if sum(wgrad_groups) != self.layers or any(g <= 0 or g > 12 for g in wgrad_groups):
                raise ValueError(f"wgrad_groups {wgrad_groups} must split {self.layers} layers into groups of 1..12")
            hi = self.layers
            for gi, g in enumerate(wgrad_groups):
                lo = hi - g
                for l in reversed(range(lo, hi)):
                    check(lib().spn_text_bwd_layer_deferred(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts),
                                                            _p(self.grads), l, ws, n, _stream()), "text_bwd_layer_deferred")
                if embed_early and gi == len(wgrad_groups) - 1:
                    check(lib().spn_text_bwd_tail(C.byref(cfg), _p(ids), _p(self._acts), _p(self.grads), ws, n, _stream()),
                          "text_bwd_tail")
                    on_span_ready(*spans[-1])
                check(lib().spn_text_bwd_wgrad(C.byref(cfg), _p(self._acts), _p(self.grads), lo, hi, ws, n, _stream()),
                      "text_bwd_wgrad")
                for l in reversed(range(lo, hi)):
                    on_span_ready(*spans[1 + (self.layers - 1 - l)])
                hi = lo
            if embed_early:
                return self.grads
        check(lib().spn_text_bwd_tail(C.byref(cfg), _p(ids), _p(self._acts), _p(self.grads), ws, n, _stream()),
              "text_bwd_tail")
        on_span_ready(*spans[-1])
        return self.grads
