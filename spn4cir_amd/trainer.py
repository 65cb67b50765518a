"""Fast-path stage-2 trainer: one `step()` = the loop body of clip4cir/train_negplus.py:107-123.

tokens-on-device -> text tower fwd -> combiner + L2-normalise -> bank InfoNCE -> backward ->
(gradient all-reduce) -> AdamW -> bf16 weight refresh, with no host synchronisation inside a step
(the reference syncs on loss.item() every step, train_negplus.py:114; here the loss stays on
the device and is read by the caller when it wants it).  bf16 needs no loss scaling, so the
GradScaler of the reference (train_negplus.py:84) degenerates to scale = 1; the inv_scale /
found_inf plumbing of spn_adamw_step is still exercised by the autograd path in models.py.
"""
import torch

from . import ops
from . import distributed as _dp
from .distributed import BankLossDP, GradBucketReducer, SparseRowReducer, _world, shard_range


def wgrad_groups(layers, world):
    """How the backward pass batches the weight gradients of its blocks (TextTower.backward_phased): one process takes
    all of them in one grouped launch (<= 12 blocks per launch); data-parallel ranks use three groups (7 + 3 + 2 of 12
    blocks), so that each group's gradient all-reduce runs under the backward of the following ones (partial rounds of
    256 CUs are split over the reduction by the kernel).
    SPN_WGRAD_GROUPS="a,b,.." overrides; "0" = no deferral (one grouped launch per block)."""
    from ._lib import env as lib_env          # the library's own snapshot of the environment (one source for both sides)
    env = lib_env("SPN_WGRAD_GROUPS")
    if env is not None:
        return None if env.strip() in ("", "0") else [int(x) for x in env.split(",")]
    if world > 1 and 4 <= layers <= 24:
        # three groups: the all-reduce of a group runs under the data path and weight gradients of the next ones, the
        # embedding gradients' (Stage2Trainer: reported before the LAST group's launch) under that launch, and only the
        # small last group's own all-reduce is left for the AdamW update of everything else to cover
        first = min(12, (7 * layers + 11) // 12)
        last = max(1, min(layers - first, (layers + 3) // 6))
        mid = layers - first - last
        groups = [first] + ([mid] if 0 < mid <= 12 else ([12, mid - 12] if mid > 12 else [])) + [last]
        return groups
    groups, left = [], layers
    while left > 0:
        groups.append(min(12, left))
        left -= groups[-1]
    return groups


class Stage2Trainer:
    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01, group=None,
                 bank_mode="replicated", check_finite=False, pack=True, grad_comm_dtype="fp32", grad_comm_algo=None,
                 optim="replicated"):
        self.model, self.tower = model, model.tower
        # pack (default): when step() gets the ids on the host too (ids_host=), the text tower computes only the rows up to each
        # caption's EOT token - same features bit for bit, same loss and gradients.  The prefix sums are built on the host and go
        # up through a pinned staging buffer without synchronising the stream.
        self.pack = bool(pack)
        self._cu_pinned = None
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.group = group
        self.world, self.rank = _world(group)
        self.loss_dp = BankLossDP(ops, group, bank_mode if (self.world > 1 or _dp._FORCE) else "replicated")
        self.m = torch.zeros_like(self.tower.params)
        self.v = torch.zeros_like(self.tower.params)
        self.step_count = 0
        self.check_finite = check_finite
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=self.tower.device)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=self.tower.device)   # applied steps (check_finite mode)
        # grad_comm_dtype="bf16": the dense gradient buckets cross the links as bf16 (all-to-all + fp32 sum in rank order +
        # all-gather; distributed.GradBucketReducer) - half the bytes, replicas still bit-identical
        # grad_comm_algo="direct" (fp32): the same exchange with fp32 payloads instead of RCCL's ring all-reduce
        # optim="sharded" (data parallel; ZeRO-1 shape): a bucket's reduced gradient chunk is consumed where the first half of the
        # direct exchange leaves it - the owner runs AdamW on its 1 / G slice and the second collective all-gathers the updated
        # fp32 masters instead of the reduced gradients (same link bytes; the 28 B / parameter optimizer stream shrinks by G on
        # every rank).  train_negplus.py:77-84,121-123 has one optimizer on one device; replicas stay bit-identical because
        # everybody takes the owner's bits.  m / v are kept full size but only the owned chunks are live.  The token embedding
        # (exchanged as touched rows) and buckets the direct exchange cannot split are still updated on every rank.
        if optim not in ("replicated", "sharded"):
            raise ValueError(optim)
        if optim == "sharded" and check_finite:
            raise ValueError("optim='sharded' has no skipped-step path (check_finite); bf16 training needs no loss scaling")
        self.optim = optim if (self.world > 1 or _dp._FORCE) else "replicated"
        if self.optim == "sharded":
            self.reducer = GradBucketReducer(self.tower.grads, group, comm_dtype=grad_comm_dtype, algo="direct",
                                             shard_update=self._update_range, flat_params=self.tower.params)
        else:
            self.reducer = GradBucketReducer(self.tower.grads, group, comm_dtype=grad_comm_dtype, algo=grad_comm_algo)
        # token-embedding gradients: exchanged as touched rows when the caller also hands the ids on the host (step(ids_host=))
        self.sparse_embed = SparseRowReducer(group) if (self.world > 1 or _dp._FORCE) else None
        self._bank = None
        self._m_begin, self._M_total = 0, 0
        self._refer = None

    def _update_range(self, lo, hi, grad=None):
        """AdamW on the flat range [lo, hi) with `grad` (default: the flat gradient's own slice) at the current step count."""
        t = self.tower
        g = t.grads[lo:hi] if grad is None else grad
        ops.adamw_step(t.params[lo:hi], g, self.m[lo:hi], self.v[lo:hi], self.step_count, self.lr, self.betas, self.eps, self.wd,
                       1.0, None)

    def set_banks(self, refer_bank, target_bank, bank_dtype="bf16"):
        """refer_bank fp32 [N, D] raw features; target_bank fp32 [M, D] L2-normalised rows.
        In "sharded" mode only this rank's contiguous row range is kept on the device.
        bank_dtype "fp8": e4m3 + per-row scale storage (BASELINE config 5) instead of bf16."""
        dev = self.tower.device
        self._refer = refer_bank.to(dev, torch.float32).contiguous()
        M = target_bank.shape[0]
        self._M_total = M
        if self.loss_dp.mode == "sharded" and self.world > 1:
            b, e = shard_range(M, self.world, self.rank)
            self._m_begin = b
            self._bank = ops.prepare_bank(target_bank[b:e].to(dev, torch.float32).contiguous(), bank_dtype)
        else:
            self._m_begin = 0
            self._bank = ops.prepare_bank(target_bank.to(dev, torch.float32), bank_dtype)

    def step(self, ids, refer_idx, labels, cu_seqlens=None, total_rows=0, ids_host=None):
        """ids int32 [B_local, L], refer_idx / labels int64 [B_local] (device). Returns the global mean
        loss as a 1-element device tensor.  cu_seqlens / total_rows (TextTower.cu_seqlens of the host ids,
        uploaded) switch the text tower to its packed mode: same result, only live rows computed; with `ids_host` and the
        trainer's pack=True (default) they are derived here.
        ids_host (data parallel only): the same ids as a CPU tensor - the token-embedding gradient is then summed over the
        ranks as touched rows (SparseRowReducer) instead of as a dense 152 MB all-reduce."""
        t = self.tower
        if cu_seqlens is None and self.pack and ids_host is not None:
            cu_host, total_rows = t.cu_seqlens(ids_host)
            if self._cu_pinned is None or self._cu_pinned.numel() != cu_host.numel():
                self._cu_pinned = torch.empty(cu_host.numel(), dtype=torch.int32).pin_memory()
                self._cu_dev = torch.empty(cu_host.numel(), dtype=torch.int32, device=t.device)
                self._cu_free = None
            if self._cu_free is not None:
                self._cu_free.synchronize()               # the previous step's upload has left the staging buffer (long ago)
            self._cu_pinned.copy_(cu_host)
            self._cu_dev.copy_(self._cu_pinned, non_blocking=True)
            self._cu_free = torch.cuda.Event()
            self._cu_free.record()
            cu_seqlens = self._cu_dev
        sparse = self.sparse_embed is not None and ids_host is not None and not self.check_finite
        if sparse:
            self.sparse_embed.plan(ids_host)              # host-side collectives only; before any device work of the step
        feats = t.forward(ids, cu_seqlens, total_rows)
        q, qb, inv = ops.combine_l2norm_fwd(self._refer, refer_idx, feats)
        ctx = self.loss_dp.forward(qb, labels, self._bank, self._m_begin, self._M_total, 1.0 / self.model.tau,
                                   self.model.label_smoothing)
        dq = self.loss_dp.backward(ctx)
        dtext = ops.combine_l2norm_bwd(q, inv, dq[:, :t.embed_dim].contiguous())
        # Data parallel: the embedding gradients (the head of the flat buffer, 38 M of the 124 M parameters) need the data
        # path of block 0, i.e. the end of backward - but not the deferred weight gradients: they are computed and their
        # all-reduce is started BEFORE the last group's grouped launch (backward_phased(embed_early=True)) and runs under
        # it.  What is left in flight at the end is the last group's own (small) bucket: it runs under the AdamW update
        # of everything else (HBM-bound vs link-bound), then that range is updated.
        spans = t.layer_spans()
        groups = wgrad_groups(t.layers, self.world)
        split = self.world > 1 and not self.check_finite and groups is not None
        # flat range of the last group's blocks (blocks 0 .. groups[-1]-1): contiguous behind the embeddings
        keep = (spans[-1][1], spans[-1 - groups[-1]][1]) if split else None

        tok_elems = t.vocab * t.width                     # flat range of token_embedding.weight: [0, V * W)
        tok_grad = t.grads[:tok_elems].view(t.vocab, t.width)

        def on_span(start, end):
            if start == 0 and split:
                self.reducer.flush()                      # everything before the embeddings goes out as its own bucket(s)
            if start == 0 and sparse:
                self.sparse_embed.start(tok_grad)         # touched rows of the token embedding, one all-gather
                self.reducer.on_span_ready(tok_elems, end)    # positional embedding: dense, tiny
            else:
                self.reducer.on_span_ready(start, end)
            if start == 0 and split:
                self.reducer.flush()                      # ... and so do the embeddings, now

        if self.optim == "sharded":
            self.step_count += 1                          # the owners' updates run inside the bucket exchanges below
            t.backward_phased(dtext, on_span, groups, embed_early=split)
            inflight = self.reducer.finish_unsharded()    # fallback all-reduces are in; the sharded exchanges keep running
            if sparse:
                self.sparse_embed.finish(tok_grad)
            # what no owner updated (the token embedding's touched-row exchange, fallback buckets): on every rank, under the
            # last exchanges
            for lo, hi in self.reducer.complement_spans(t.params.numel()):
                self._update_range(lo, hi)
            for w in inflight:
                w.wait()
            t.refresh()
            return ctx["loss"]
        t.backward_phased(dtext, on_span, groups, embed_early=split)
        pending = self.reducer.finish(keep_span=keep)
        if sparse:
            self.sparse_embed.finish(tok_grad)
        self.step_count += 1
        found = None
        if self.check_finite:
            self.found_inf.zero_()
            ops.grad_check_finite(t.grads, self.found_inf)
            found = self.found_inf
        if split and pending:
            lo, hi = keep

            def upd(a, b):
                ops.adamw_step(t.params[a:b], t.grads[a:b], self.m[a:b], self.v[a:b], self.step_count, self.lr, self.betas,
                               self.eps, self.wd, 1.0, None)
            upd(0, lo)
            upd(hi, t.params.numel())
            for w in pending:
                w.wait()
            upd(lo, hi)
        elif found is not None:
            # a skipped step must not advance the bias correction (GradScaler.step semantics): the count lives on the device
            ops.adamw_tick(self.step_dev, found)
            ops.adamw_step_dev(t.params, t.grads, self.m, self.v, self.step_dev, self.lr, self.betas, self.eps, self.wd,
                               None, found)
        else:
            ops.adamw_step(t.params, t.grads, self.m, self.v, self.step_count, self.lr, self.betas, self.eps, self.wd,
                           1.0, None)
        t.refresh()
        return ctx["loss"]
