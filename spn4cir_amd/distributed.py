"""Data-parallel logic of the stage-2 step (one process per GPU, torch.distributed over RCCL).

The reference is single-GPU (no collective on its hot path); this module is
the MI355X-native multi-GPU design of SURVEY.md section 8(e):

* triplets are sharded across ranks; the text tower is replicated and its flat gradient is
  all-reduced in per-layer buckets as soon as each layer's backward has been enqueued;
* bank loss, two modes
    "replicated": every rank scores its own rows against the whole bank - no data-path exchange;
    "sharded"   : RCCL all-gather of the (bf16) query embeddings, every rank scores ALL rows
                  against its 1/G bank shard, all-gather of the [B,4] softmax statistics, then
                  reduce-scatter of the partial dq back to the row owners.  Per-GPU HBM traffic on
                  the bank drops by G; this is the variant north_star names.

The math is backend-agnostic: `ops` is any object with bank_stats_fwd / bank_loss_finalize /
bank_grad_q (the HIP ops in production; tests inject a CPU implementation over gloo).
"""
import os
import weakref

import torch
import torch.distributed as dist

# SPN_DP_FORCE_COLLECTIVES=1: issue every collective even in a 1-rank group, so the RCCL call pattern of the
# N-GPU path (dtypes, in-place bucket slices, async handles, stream ordering) can be exercised on a 1-GPU box.
_FORCE = os.environ.get("SPN_DP_FORCE_COLLECTIVES") == "1"


def _skip(world):
    return world == 1 and not (_FORCE and dist.is_available() and dist.is_initialized())


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def shard_range(total, world, rank):
    """Contiguous, balanced [begin, end) split (remainder to the first ranks)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def all_gather_cat(x, group=None, cache=None):
    """Concatenation of every rank's x along dim 0 (equal shapes on all ranks), gathered by ONE collective straight into a
    [world * n, ...] buffer (all_gather_into_tensor: no per-rank temporaries, no torch.cat pass).  `cache` (a dict owned by
    the caller) keeps that buffer per (shape, dtype, device) for the caller's next call with the same key - only a caller
    that consumes the result before its next call may pass one (BankLossDP: one forward in flight per instance); without
    it every call gets a fresh buffer."""
    world, _ = _world(group)
    if _skip(world):
        return x
    x = x.contiguous()
    key = (tuple(x.shape), x.dtype, str(x.device))
    out = cache.get(key) if cache is not None else None
    if out is None:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if cache is not None:
            cache[key] = out
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def _mean_over_ranks(mean_local, group=None):
    """Global mean loss from the ranks' local means (equal local batch sizes): ONE collective and no elementwise kernel on the
    device path - RCCL averages in the all-reduce itself (ReduceOp.AVG); gloo (CPU tests, two ranks sharing a GPU) sums and
    divides.  The local mean is a kernel output (spn_bank_loss_finalize / spn_bank_step), never a torch reduction."""
    out = mean_local                                   # a fresh kernel output of this call: reduced in place
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(out, op=dist.ReduceOp.AVG, group=group)
        return out
    dist.all_reduce(out, group=group)
    return out / dist.get_world_size(group)


def reduce_scatter_rows(x, group=None):
    """x [world * B_local, ...] summed over ranks; this rank keeps its B_local rows."""
    world, rank = _world(group)
    if _skip(world):
        return x
    n = x.shape[0] // world
    if dist.get_backend(group) == "gloo":        # gloo has no reduce_scatter
        y = x.contiguous().clone()
        dist.all_reduce(y, group=group)
        return y[rank * n:(rank + 1) * n].contiguous()
    out = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.reduce_scatter_tensor(out, x.contiguous(), group=group)
    return out


class _CacheToken:
    """Ownership marker of BankLossDP's gather buffers (weak-referenced: a dropped ctx releases them by itself)."""


class BankLossDP:
    def __init__(self, ops, group=None, mode="sharded"):
        if mode not in ("replicated", "sharded"):
            raise ValueError(mode)
        self.ops, self.group, self.mode = ops, group, mode
        self.world, self.rank = _world(group)
        # gather buffers of the sharded mode, private to this instance and reused step after step.  They back ctx['q'] /
        # ctx['labels'] until backward(ctx) has run: a second forward before that (gradient accumulation, an eval forward
        # between forward and backward) gets fresh buffers instead of overwriting the pending step's queries.
        self._bufs = {}
        self._cache_owner = None          # weak reference to the token of the ctx that currently holds self._bufs

    def forward(self, qb_local, labels_local, bank, m_begin, M_total, inv_tau, label_smoothing=0.0, need_grad=True):
        """qb_local [B_local, Dp] bf16 L2-normalised queries, labels_local [B_local] int64 GLOBAL bank
        rows, `bank` = full bank (replicated) or this rank's shard starting at global row m_begin.
        Returns a ctx dict; ctx['loss'] is the GLOBAL mean loss (identical on every rank).
        need_grad=False (evaluation, loss reporting): no backward call will follow - the one-call step, which also forms dq and
        allocates its scratch, is not taken."""
        ops = self.ops

        def stats_fwd(q, labels):
            # the HIP ops keep the logits of small per-call batches for the backward pass (ops.bank_logits_buffer)
            mk = getattr(ops, "bank_logits_buffer", None)
            save = mk(q.shape[0], bank.shape[0], q.device) if mk is not None else None
            if save is None:
                return ops.bank_stats_fwd(q, bank, labels, inv_tau, m_begin), None
            return ops.bank_stats_fwd(q, bank, labels, inv_tau, m_begin, save=save), save

        if self.mode == "replicated" or _skip(self.world):
            # the whole bank on this rank, no label smoothing: forward and backward w.r.t. q in two launches (spn_bank_step)
            step_ok = getattr(ops, "bank_step_ok", None)
            if (need_grad and step_ok is not None and label_smoothing == 0.0 and m_begin == 0 and bank.shape[0] == M_total
                    and step_ok(qb_local.shape[0], M_total, qb_local.shape[1], bank)):
                save = ops.bank_logits_buffer(qb_local.shape[0], M_total, qb_local.device)
                if save is not None:
                    B_global = qb_local.shape[0] * self.world
                    lse, row, mean, dq = ops.bank_step(qb_local, bank, labels_local, inv_tau, 1.0 / B_global, save)
                    loss = mean if _skip(self.world) else _mean_over_ranks(mean, self.group)      # reporting only
                    return dict(q=qb_local, labels=labels_local, lse=lse, loss=loss, bank=bank, m_begin=m_begin,
                                M_total=M_total, inv_tau=inv_tau, ls=label_smoothing, B_global=B_global, gathered=False,
                                saved=None, dq=dq)
            stats, saved = stats_fwd(qb_local, labels_local)
            lse, row, mean = ops.bank_loss_finalize(stats, M_total, label_smoothing)
            loss = mean if _skip(self.world) else _mean_over_ranks(mean, self.group)              # reporting only
            B_global = qb_local.shape[0] * self.world
            return dict(q=qb_local, labels=labels_local, lse=lse, loss=loss, bank=bank, m_begin=m_begin,
                        M_total=M_total, inv_tau=inv_tau, ls=label_smoothing, B_global=B_global, gathered=False, saved=saved)
        # the cached gather buffers belong to ONE ctx at a time: free again when that ctx's backward has run or the ctx was
        # dropped without one (an evaluation forward, an exception) - its token is then gone and the weak reference dead
        token = None
        cache = None
        if self._cache_owner is None or self._cache_owner() is None:
            token = _CacheToken()
            self._cache_owner = weakref.ref(token)
            cache = self._bufs
        q_all = all_gather_cat(qb_local, self.group, cache)
        labels_all = all_gather_cat(labels_local, self.group, cache)
        stats, saved = stats_fwd(q_all, labels_all)                                     # [B, 4] over my shard
        stats_all = all_gather_cat(stats.unsqueeze(0), self.group, cache)               # [G, B, 4]
        lse, row, mean = ops.bank_loss_finalize(stats_all, M_total, label_smoothing)    # identical everywhere
        return dict(q=q_all, labels=labels_all, lse=lse, loss=mean, bank=bank, m_begin=m_begin, M_total=M_total,
                    inv_tau=inv_tau, ls=label_smoothing, B_global=q_all.shape[0], gathered=True, saved=saved, cache_token=token)

    def backward(self, ctx, loss_scale=1.0):
        """-> d(loss_scale * global mean loss)/d(q_local)  [B_local, Dp] fp32"""
        if ctx.get("dq") is not None:                    # bank_step already produced d(mean loss)/dq
            return ctx["dq"] if loss_scale == 1.0 else ctx["dq"] * loss_scale
        gs = loss_scale / ctx["B_global"]
        kw = {"saved": ctx["saved"]} if ctx.get("saved") is not None else {}
        dq = self.ops.bank_grad_q(ctx["q"], ctx["bank"], ctx["labels"], ctx["inv_tau"], ctx["lse"], gs,
                                  M_total=ctx["M_total"], label_smoothing=ctx["ls"], m_begin=ctx["m_begin"], **kw)
        if ctx["gathered"]:
            dq = reduce_scatter_rows(dq, self.group)
            if ctx.get("cache_token") is not None and self._cache_owner is not None and self._cache_owner() is ctx["cache_token"]:
                self._cache_owner = None                 # this step is done with the gather buffers
            ctx["cache_token"] = None
        return dq


class SparseRowReducer:
    """Sum over ranks of a [V, W] gradient whose non-zero rows every rank can name on the HOST before the step runs: the
    token-embedding gradient (38 M of ViT-L/14's 124 M text parameters, 152 MB in fp32) is zero outside the rows of the
    token ids of the rank's own captions - a few thousand of 49 408.  A dense all-reduce of it is a third of all gradient
    bytes and, being final only when backward ends, the one bucket that cannot hide behind it.  Here instead:

      plan(ids_host)   host only, before any device work of the step is enqueued: unique ids of this rank, exchanged with
                       the other ranks over a CPU (gloo) group - two tiny collectives, no device synchronisation, the
                       device is still busy with the previous step;
      start(grad)      behind the kernels that finish the gradient: this rank's U rows are gathered into a [cap, W] send
                       buffer (cap = the largest U of any rank) and ONE all-gather moves them (async);
      finish(grad)     wait; the other ranks' rows are added into this rank's gradient (index_add_).

    Per rank (G - 1) * cap * W * 4 bytes arrive (G = 8, cap = 5.3 k, W = 768: 114 MB) against the 2 (G - 1) / G * 152 MB
    = 266 MB a ring all-reduce of the dense matrix moves; with real captions (a vocabulary of a few thousand words) the
    gap is far larger.  Every rank sums the G contributions of a row in the SAME order (rank 0, 1, .. G-1, its own
    included, starting from zero), so the replicas' gradients stay bit-identical for any G (fp32 addition does not
    associate: adding "the others" onto the own value would give rank-dependent last bits from G = 3 on)."""

    _host_groups = {}            # one CPU (gloo) group per process and device group, shared by every reducer

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = _world(group)
        self.host_group = None   # created on the first plan() call (a collective: every rank plans in the same step)
        self._plan = None
        self._work = None

    def _host(self):
        if self.host_group is None:
            if dist.get_backend(self.group) == "gloo":
                self.host_group = self.group
            else:
                key = id(self.group)
                if key not in SparseRowReducer._host_groups:
                    SparseRowReducer._host_groups[key] = dist.new_group(backend="gloo")
                self.host_group = SparseRowReducer._host_groups[key]
        return self.host_group

    def plan(self, ids_host):
        if _skip(self.world):
            self._plan = None
            return
        self._host()
        uniq = torch.unique(ids_host.reshape(-1).to(torch.int64))
        n = torch.tensor([uniq.numel()], dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(counts, n, group=self.host_group)
        counts = [int(c) for c in counts]
        cap = max(1, max(counts))
        mine = torch.zeros(cap, dtype=torch.int64)
        mine[:uniq.numel()] = uniq
        rows_all = [torch.zeros(cap, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(rows_all, mine, group=self.host_group)
        self._plan = dict(counts=counts, cap=cap, rows=rows_all, uniq=uniq)

    def start(self, grad2d):
        """grad2d [V, W]: this rank's (final) gradient.  Enqueues the gather of its rows and the all-gather."""
        pl = self._plan
        if pl is None:
            return
        dev = grad2d.device
        if "rows_dev" not in pl:
            pl["rows_dev"] = [r[:c].to(dev, non_blocking=True) for r, c in zip(pl["rows"], pl["counts"])]
        send = torch.zeros(pl["cap"], grad2d.shape[1], dtype=grad2d.dtype, device=dev)
        u = pl["counts"][self.rank]
        if u:
            send[:u] = grad2d.index_select(0, pl["rows_dev"][self.rank])
        recv = torch.empty(self.world * pl["cap"], grad2d.shape[1], dtype=grad2d.dtype, device=dev)
        if dist.get_backend(self.group) == "gloo":
            parts = list(recv.view(self.world, pl["cap"], grad2d.shape[1]).unbind(0))
            self._work = dist.all_gather(parts, send, group=self.group, async_op=True)
        else:
            self._work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=True)
        pl["recv"] = recv

    def finish(self, grad2d):
        pl = self._plan
        if pl is None or self._work is None:
            return
        self._work.wait()
        self._work = None
        cap = pl["cap"]
        u = pl["counts"][self.rank]
        if u:
            grad2d.index_fill_(0, pl["rows_dev"][self.rank], 0.0)    # own rows come back through recv, in rank order
        for r in range(self.world):
            c = pl["counts"][r]
            if c:      # the ids of one rank are unique: no two addends of a call hit the same element
                grad2d.index_add_(0, pl["rows_dev"][r], pl["recv"][r * cap:r * cap + c])
        self._plan = None


class _HipExchangeKernels:
    """The three elementwise passes of the bf16 gradient exchange on the library's kernels (spn_cast_f32_bf16,
    spn_sum_ranks_bf16, spn_cast_bf16_f32), on torch's current stream."""
    ALIGN = 4        # elements: the kernels move 16-byte vectors, a slice must start on a 4-element boundary of the flat buffer

    @staticmethod
    def to_bf16(src_f32, dst_bf16):
        from . import ops
        ops.check(ops.lib().spn_cast_f32_bf16(ops._p(src_f32), ops._p(dst_bf16), src_f32.numel(), ops._stream()), "cast_f32_bf16")

    @staticmethod
    def sum_ranks(chunks_bf16, world, out_bf16):
        from . import ops
        ops.check(ops.lib().spn_sum_ranks_bf16(ops._p(chunks_bf16), world, out_bf16.numel(), ops._p(out_bf16), ops._stream()),
                  "sum_ranks_bf16")

    @staticmethod
    def to_f32(src_bf16, dst_f32):
        from . import ops
        ops.check(ops.lib().spn_cast_bf16_f32(ops._p(src_bf16), ops._p(dst_f32), dst_f32.numel(), ops._stream()), "cast_bf16_f32")

    @staticmethod
    def sum_ranks_f32(chunks_f32, world, out_f32):
        from . import ops
        ops.check(ops.lib().spn_sum_ranks_f32(ops._p(chunks_f32), world, out_f32.numel(), ops._p(out_f32), ops._stream()),
                  "sum_ranks_f32")


class _Bf16Work:
    """Handle of one direct bucket exchange: wait() = the second collective has landed, the summed slice is in the flat fp32
    gradient (cast back from `full` for the bf16 flavour; gathered in place for fp32), and the caller's current stream is ordered
    behind all of it."""

    def __init__(self, reducer, s, e, full, work, comm):
        self.r, self.s, self.e, self.full, self.work, self.comm = reducer, s, e, full, work, comm

    def wait(self):
        r = self.r
        with r._on(self.comm):
            self.work.wait()
            if self.full is not None:
                r.kernels.to_f32(self.full[:self.e - self.s], r.flat[self.s:self.e])
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        self.full = None


class GradBucketReducer:
    """Sum the flat gradient over ranks in buckets, overlapped with the rest of backward.

    `on_span_ready(start, end)` is called by TextTower.backward_phased as soon as the kernels that
    finalise flat[start:end] are enqueued; torch's RCCL stream waits for exactly that point of the
    compute stream and runs the all-reduce concurrently with the following layers.  Small spans are
    merged until `bucket_elems` is reached (xGMI all-reduce wants >= tens of MB per call).

    comm_dtype="bf16" (optional; default fp32 = one all-reduce per bucket): half the bytes on the links (247 MB
    instead of 494 MB per step for ViT-L/14's text tower) with fp32 accumulation and bit-identical replicas.  Per bucket:
      cast the slice to bf16 [G x m] -> all-to-all (rank j receives every rank's chunk j) -> sum the G chunks in fp32 IN RANK
      ORDER, round once to bf16 (spn_sum_ranks_bf16) -> all-gather of the reduced chunks -> cast back into the flat fp32 gradient.
    Every replica - the chunk's owner included - takes the all-gathered bf16 value, so the replicas stay bit-identical; what the
    gradient loses is one bf16 rounding of each rank's contribution and one of the sum (the bf16-operand weight-gradient GEMMs
    that produced it carry rounding of the same size).  The passes run on a dedicated stream between the two collectives; the
    host never blocks.  `kernels` = the elementwise passes (the HIP kernels by default; CPU tests inject torch ones)."""

    def __init__(self, flat_grads, group=None, bucket_elems=8 << 20, comm_dtype="fp32", kernels=None, algo=None,
                 shard_update=None, flat_params=None):
        if comm_dtype not in ("fp32", "bf16"):
            raise ValueError(comm_dtype)
        # Sharded optimizer step (ZeRO-1 shape; Stage2Trainer(optim="sharded")): after the first half of the direct exchange
        # (all-to-all + rank-order sum) rank r holds the REDUCED chunk r of a bucket.  Instead of all-gathering the reduced
        # gradients and letting every rank update all parameters, `shard_update(lo, hi, grad_f32)` updates flat_params[lo:hi]
        # on the owner only, and the second collective all-gathers the updated fp32 MASTERS into flat_params[s:e] - the same
        # link bytes, 1 / G of the optimizer's HBM traffic per rank.  Every replica takes the owner's bits: identical by
        # construction.  Buckets the direct exchange cannot take (length not a multiple of 4 * world, unaligned start) fall
        # back to an all-reduce; `sharded_spans` lists what WAS updated, the caller updates the complement itself.
        if shard_update is not None and flat_params is None:
            raise ValueError("shard_update needs flat_params")
        self.shard_update, self.flat_params = shard_update, flat_params
        self.sharded_spans = []
        if shard_update is not None:
            algo = "direct"
        # algo "ring" = one RCCL all-reduce per bucket (fp32 only); "direct" = all-to-all + rank-order sum + all-gather: every
        # rank exchanges one S / G chunk with every other rank over its own xGMI link, all links at once (SURVEY section 5's
        # one-shot reduce-scatter + all-gather, from RCCL's point-to-point collectives).  bf16 implies direct.
        algo = algo or ("direct" if comm_dtype == "bf16" else "ring")
        # algo "none": MEASUREMENT ONLY (bench.py's overlap figure) - buckets are not exchanged at all, the replicas drift apart
        if algo not in ("ring", "direct", "none") or (comm_dtype == "bf16" and algo != "direct"):
            raise ValueError((comm_dtype, algo))
        self.algo = algo
        self.flat, self.group, self.bucket_elems = flat_grads, group, bucket_elems
        self.world, self.rank = _world(group)
        self.comm_dtype = comm_dtype
        self.kernels = kernels or _HipExchangeKernels
        self._align = getattr(self.kernels, "ALIGN", 1)
        self._comm_stream = None
        self._pending = []      # [(start, end)] contiguous-or-not spans waiting for a bucket
        self._works = []

    def _on(self, stream):
        import contextlib
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def _exchange_bf16(self, s, e):
        """-> handle whose wait() completes flat[s:e] = sum over ranks (see the class docstring)."""
        G, n = self.world, e - s
        m = ((n + G - 1) // G + 7) // 8 * 8
        comm = None
        if self.flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.flat.device)
            comm = self._comm_stream
            comm.wait_stream(torch.cuda.current_stream())          # the slice is final at this point of the compute stream
        with self._on(comm):
            # padding behind element n (the tail of the last chunk): zeros - it is summed and gathered like everything else
            # (never read back, but uninitialised bf16 can be NaN, and a NaN that travels is a trap for the next change here)
            send = torch.empty(G * m, dtype=torch.bfloat16, device=self.flat.device)
            if G * m > n:
                send[n:].zero_()
            self.kernels.to_bf16(self.flat[s:e], send[:n])
            recv = torch.empty(G * m, dtype=torch.bfloat16, device=self.flat.device)
            w1 = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
            w1.wait()                                              # stream-ordered on the communication stream, not the host
            red = torch.empty(m, dtype=torch.bfloat16, device=self.flat.device)
            self.kernels.sum_ranks(recv, G, red)
            full = torch.empty(G * m, dtype=torch.bfloat16, device=self.flat.device)
            w2 = dist.all_gather_into_tensor(full, red, group=self.group, async_op=True)
        return _Bf16Work(self, s, e, full, w2, comm)

    def _exchange_f32(self, s, e):
        """direct fp32 exchange, in place: needs (e - s) % (4 * world) == 0 (true for every span of the towers here); None
        otherwise (the caller falls back to the ring all-reduce for that bucket)."""
        G, n = self.world, e - s
        if n % (4 * G):
            return None
        m = n // G
        comm = None
        if self.flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.flat.device)
            comm = self._comm_stream
            comm.wait_stream(torch.cuda.current_stream())
        with self._on(comm):
            recv = torch.empty(n, dtype=torch.float32, device=self.flat.device)
            w1 = dist.all_to_all_single(recv, self.flat[s:e], group=self.group, async_op=True)
            w1.wait()
            red = torch.empty(m, dtype=torch.float32, device=self.flat.device)
            self.kernels.sum_ranks_f32(recv, G, red)
            w2 = dist.all_gather_into_tensor(self.flat[s:e], red, group=self.group, async_op=True)
        return _Bf16Work(self, s, e, None, w2, comm)

    def _exchange_sharded(self, s, e):
        """all-to-all (fp32 or bf16 payload) -> rank-order sum -> optimizer on the owned chunk -> all-gather of the updated
        masters.  None when the bucket does not split into `world` 16-byte aligned chunks (the caller all-reduces it and the
        trainer updates it on every rank)."""
        G, n = self.world, e - s
        if n % (4 * G) or s % self._align:
            return None
        m = n // G
        dev = self.flat.device
        comm = None
        if self.flat.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=dev)
            comm = self._comm_stream
            comm.wait_stream(torch.cuda.current_stream())          # gradients of [s, e) final; nobody reads its parameters any more
        with self._on(comm):
            red = torch.empty(m, dtype=torch.float32, device=dev)
            if self.comm_dtype == "bf16":
                send = torch.empty(n, dtype=torch.bfloat16, device=dev)
                self.kernels.to_bf16(self.flat[s:e], send)
                recv = torch.empty(n, dtype=torch.bfloat16, device=dev)
                dist.all_to_all_single(recv, send, group=self.group, async_op=True).wait()
                red_b = torch.empty(m, dtype=torch.bfloat16, device=dev)
                self.kernels.sum_ranks(recv, G, red_b)
                self.kernels.to_f32(red_b, red)
            else:
                recv = torch.empty(n, dtype=torch.float32, device=dev)
                dist.all_to_all_single(recv, self.flat[s:e], group=self.group, async_op=True).wait()
                self.kernels.sum_ranks_f32(recv, G, red)
            lo = s + self.rank * m
            self.shard_update(lo, lo + m, red)
            own = self.flat_params[lo:lo + m].clone()              # (gloo has no in-place all-gather; S / G elements)
            w2 = dist.all_gather_into_tensor(self.flat_params[s:e], own, group=self.group, async_op=True)
        self.sharded_spans.append((s, e))
        return _Bf16Work(self, s, e, None, w2, comm)

    def complement_spans(self, total):
        """[0, total) minus the spans the sharded update handled since the last call: what the caller updates on every rank."""
        out, pos = [], 0
        for s, e in sorted(self.sharded_spans):
            if s > pos:
                out.append((pos, s))
            pos = max(pos, e)
        if pos < total:
            out.append((pos, total))
        self.sharded_spans = []
        return out

    def _flush(self):
        if not self._pending:
            return
        # merge adjacent spans (backward walks the flat buffer from the end, so they usually abut)
        spans = sorted(self._pending)
        merged = [list(spans[0])]
        for s, e in spans[1:]:
            if s == merged[-1][1]:
                merged[-1][1] = e
            else:
                merged.append([s, e])
        for s, e in merged:
            work = None
            if self.shard_update is not None:
                work = self._exchange_sharded(s, e)
            elif self.comm_dtype == "bf16":
                # the cast / sum kernels move 16-byte vectors: a span that does not start on a 4-element boundary (no tower
                # here has one) takes the ring all-reduce instead of failing inside wait()
                work = self._exchange_bf16(s, e) if s % self._align == 0 else None
            elif self.algo == "direct":
                work = self._exchange_f32(s, e) if s % self._align == 0 else None
            if work is None:
                work = dist.all_reduce(self.flat[s:e], group=self.group, async_op=True)
            self._works.append((s, e, work))
        self._pending = []

    def on_span_ready(self, start, end):
        if _skip(self.world) or self.algo == "none":
            return
        self._pending.append((start, end))
        if sum(e - s for s, e in self._pending) >= self.bucket_elems:
            self._flush()

    def finish_unsharded(self):
        """Sharded-update mode: wait for the fallback all-reduces only; returns the handles of the sharded exchanges (their
        wait() = the gathered masters are in flat_params and the compute stream is ordered behind them)."""
        if _skip(self.world):
            return []
        self._flush()
        works, self._works = self._works, []
        kept = []
        for _, _, w in works:
            if isinstance(w, _Bf16Work):
                kept.append(w)
            else:
                w.wait()
        return kept

    def finish(self, keep_span=None):
        """Make the compute stream wait for the all-reduces.  keep_span = (start, end) leaves the all-reduces of
        exactly that flat range in flight and returns their handles: the caller overlaps them with work that does
        not touch the range (AdamW over the rest of the buffer) and then calls .wait() on each.  Handles are kept
        only when their spans lie inside keep_span AND tile it completely - decided from the recorded spans, not
        from issue order; otherwise everything is waited for and [] is returned."""
        if _skip(self.world):
            return []
        self._flush()
        works, self._works = self._works, []
        kept = []
        if keep_span is not None:
            lo, hi = keep_span
            inside = sorted((s, e) for s, e, _ in works if lo <= s and e <= hi)
            pos = lo
            for s, e in inside:
                if s != pos:
                    break
                pos = e
            if inside and pos == hi:
                kept = [w for s, e, w in works if lo <= s and e <= hi]
                works = [(s, e, w) for s, e, w in works if not (lo <= s and e <= hi)]
        for _, _, w in works:
            w.wait()            # the compute stream waits for the RCCL stream; no host sync on RCCL
        return kept

    def flush(self):
        """Close the current bucket now (its all-reduce starts behind the work enqueued so far)."""
        if not _skip(self.world):
            self._flush()
