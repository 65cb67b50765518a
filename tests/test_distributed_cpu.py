"""World-size-2 gloo tests of the data-parallel logic (spn4cir_amd/distributed.py).

The collectives and the shard / LSE-combination math run for real over gloo; the three bank ops
are injected as a CPU implementation built on the oracle (allowed in tests only), so the test
checks that the multi-rank result equals the single-process oracle loss and gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class CpuBankOps:
    """bank_stats_fwd / bank_loss_finalize / bank_grad_q with the semantics of include/spn4cir_hip.h."""

    @staticmethod
    def bank_stats_fwd(q, bank, labels, inv_tau, m_begin=0):
        logits = (q.double() @ bank.double().t()) * inv_tau
        m = logits.max(dim=1).values
        l = torch.exp(logits - m[:, None]).sum(1)
        sl = logits.sum(1)
        local = labels.long() - m_begin
        inside = (local >= 0) & (local < bank.shape[0])
        lab = torch.full_like(m, float("-inf"))
        rows = torch.arange(q.shape[0])[inside]
        lab[inside] = logits[rows, local[inside]]
        return torch.stack([m, l, sl, lab], dim=1).float()

    @staticmethod
    def bank_loss_finalize(stats, M_total, label_smoothing=0.0):
        if stats.dim() == 2:
            stats = stats.unsqueeze(0)
        s = stats.double()
        m = s[..., 0].max(dim=0).values
        l = (s[..., 1] * torch.exp(s[..., 0] - m)).sum(0)
        lse = m + torch.log(l)
        lab = s[..., 3].max(dim=0).values
        row = lse - (1 - label_smoothing) * lab - label_smoothing * s[..., 2].sum(0) / M_total
        return lse.float(), row.float(), row.mean().reshape(1).float()

    @staticmethod
    def bank_grad_q(q, bank, labels, inv_tau, row_lse, grad_scale, M_total=None, label_smoothing=0.0, m_begin=0):
        logits = (q.double() @ bank.double().t()) * inv_tau
        g = torch.exp(logits - row_lse.double()[:, None]) - label_smoothing / (M_total or bank.shape[0])
        local = labels.long() - m_begin
        inside = (local >= 0) & (local < bank.shape[0])
        rows = torch.arange(q.shape[0])[inside]
        g[rows, local[inside]] -= 1 - label_smoothing
        return ((g @ bank.double()) * (grad_scale * inv_tau)).float()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spn4cir_amd.distributed import BankLossDP, GradBucketReducer, shard_range
        g = torch.Generator().manual_seed(0)
        B, M, D, tau, eps = 12, 301, 32, 0.05, 0.1
        q = torch.nn.functional.normalize(torch.randn(B, D, generator=g))
        bank = torch.nn.functional.normalize(torch.randn(M, D, generator=g))
        labels = torch.randint(0, M, (B,), generator=g)
        bl = B // world
        ql, ll = q[rank * bl:(rank + 1) * bl].clone(), labels[rank * bl:(rank + 1) * bl].clone()
        dp = BankLossDP(CpuBankOps, None, mode)
        if mode == "sharded":
            b, e = shard_range(M, world, rank)
            ctx = dp.forward(ql, ll, bank[b:e].contiguous(), b, M, 1.0 / tau, eps)
        else:
            ctx = dp.forward(ql, ll, bank, 0, M, 1.0 / tau, eps)
        dq = dp.backward(ctx)
        ok_cache = True
        if mode == "sharded":
            # gather-buffer ownership: a forward whose ctx is dropped without backward (evaluation) must not disable the
            # cache for good, and a second forward while a step is pending must not overwrite that step's queries
            qbuf = ctx["q"]
            args = (ql, ll, bank[b:e].contiguous(), b, M, 1.0 / tau, eps)
            c1 = dp.forward(*args)                                   # backward done above: the cached buffers are free again
            ok_cache = c1["q"].data_ptr() == qbuf.data_ptr()
            c2 = dp.forward(*args)                                   # c1 still pending: fresh buffers
            ok_cache = ok_cache and c2["q"].data_ptr() != qbuf.data_ptr() and torch.equal(c1["q"], c2["q"])
            del c1, c2                                               # both dropped without backward
            c3 = dp.forward(*args, need_grad=False)
            ok_cache = ok_cache and c3["q"].data_ptr() == qbuf.data_ptr()
            dq3 = dp.backward(c3)
            ok_cache = ok_cache and torch.equal(dq3, dq) and abs(c3["loss"].item() - ctx["loss"].item()) < 1e-7
        # single-process reference
        qd = q.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy((qd @ bank.double().t()) / tau, labels, label_smoothing=eps)
        ref.backward()
        ok_loss = abs(ctx["loss"].item() - ref.item()) < 1e-5
        ok_grad = torch.allclose(dq.double(), qd.grad[rank * bl:(rank + 1) * bl], atol=1e-6, rtol=1e-4) and ok_cache
        # bucketed gradient all-reduce
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red = GradBucketReducer(flat, None, bucket_elems=300)
        for s, e in [(900, 1000), (600, 900), (250, 600), (0, 250)]:
            red.on_span_ready(s, e)
        red.finish()
        ok_red = torch.equal(flat, torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1)))
        # finish(keep_span): the kept handles are chosen by their recorded spans - here the head span [0, 250) is
        # issued FIRST, so "the last work" would be the wrong one - and only when they tile the span exactly
        want = torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1))
        flat2 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red2 = GradBucketReducer(flat2, None, bucket_elems=300)
        red2.on_span_ready(0, 250)
        red2.flush()
        for s, e in [(900, 1000), (600, 900), (250, 600)]:
            red2.on_span_ready(s, e)
        kept = red2.finish(keep_span=(0, 250))
        ok_red = ok_red and len(kept) == 1 and torch.equal(flat2[250:], want[250:])
        for w in kept:
            w.wait()
        ok_red = ok_red and torch.equal(flat2, want)
        flat3 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        red3 = GradBucketReducer(flat3, None, bucket_elems=10 ** 9)       # one bucket [0, 1000): nothing tiles [0, 250)
        for s, e in [(250, 1000), (0, 250)]:
            red3.on_span_ready(s, e)
        ok_red = ok_red and red3.finish(keep_span=(0, 250)) == [] and torch.equal(flat3, want)
        # bf16 bucket exchange: all-to-all + fp32 sum in rank order + all-gather; every rank ends with the SAME bits, equal to
        # bf16(sum_r fp32(bf16(g_r))) element by element, also for a bucket length that is no multiple of world * 8
        class TorchKernels:
            @staticmethod
            def to_bf16(src, dst):
                dst.copy_(src.to(torch.bfloat16))

            @staticmethod
            def sum_ranks(chunks, G, out_):
                out_.copy_(chunks.view(G, -1).float().sum(0).to(torch.bfloat16))

            @staticmethod
            def to_f32(src, dst):
                dst.copy_(src.float())

            @staticmethod
            def sum_ranks_f32(chunks, G, out_):
                acc = torch.zeros_like(out_)
                for r in range(G):                                   # rank order, as the kernel
                    acc += chunks.view(G, -1)[r]
                out_.copy_(acc)
        gg = torch.Generator().manual_seed(5)
        base = torch.randn(world, 1003, generator=gg)
        flat4 = base[rank].clone()
        red4 = GradBucketReducer(flat4, None, bucket_elems=400, comm_dtype="bf16", kernels=TorchKernels)
        for s, e in [(700, 1003), (301, 700), (0, 301)]:
            red4.on_span_ready(s, e)
        kept4 = red4.finish(keep_span=(0, 301))
        for w in kept4:
            w.wait()
        want4 = base.to(torch.bfloat16).float().sum(0).to(torch.bfloat16).float()
        ok_red = ok_red and len(kept4) == 1 and torch.equal(flat4, want4)
        # direct fp32 exchange (all-to-all + rank-order sum + all-gather, in place); a bucket whose length is no multiple of
        # 4 * world falls back to the ring all-reduce
        flat5 = base[rank, :1000].clone()
        red5 = GradBucketReducer(flat5, None, bucket_elems=300, kernels=TorchKernels, algo="direct")
        for s, e in [(600, 1000), (296, 600), (0, 296)]:
            red5.on_span_ready(s, e)
        red5.finish()
        want5 = torch.zeros(1000)
        for r in range(world):
            want5 += base[r, :1000]
        ok_red = ok_red and torch.equal(flat5, want5)
        flat6 = base[rank, :1003].clone()
        red6 = GradBucketReducer(flat6, None, bucket_elems=10 ** 9, kernels=TorchKernels, algo="direct")
        red6.on_span_ready(0, 1003)
        red6.finish()
        ok_red = ok_red and torch.allclose(flat6, base[:, :1003].sum(0), atol=1e-6)
        out.put((rank, ok_loss, ok_grad, ok_red, ctx["loss"].item(), ref.item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sharded", "replicated"])
def test_bank_loss_data_parallel_world2(mode):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_loss, ok_grad, ok_red, loss, ref in res:
        assert ok_loss, (mode, rank, loss, ref)
        assert ok_grad, (mode, rank)
        assert ok_red, (mode, rank)


def _sparse_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spn4cir_amd.distributed import SparseRowReducer
        V, W, B, L = 500, 24, 6, 11
        ok = True
        red = SparseRowReducer(None)
        for step in range(3):                                 # different ids every step, unequal unique counts per rank
            g = torch.Generator().manual_seed(100 * step + rank)
            ids = torch.randint(0, V if rank else 40, (B, L), generator=g, dtype=torch.int32)   # rank 0: few distinct rows
            ids[:, 0] = 7                                      # a row every rank touches
            grad = torch.zeros(V, W)
            grad.index_add_(0, ids.reshape(-1).long(), torch.randn(B * L, W, generator=g))      # non-zero only on own ids
            dense = grad.clone()
            dist.all_reduce(dense)                             # what the dense bucket would give
            red.plan(ids)
            red.start(grad)
            red.finish(grad)
            ok = ok and torch.allclose(grad, dense, atol=1e-5, rtol=1e-5)
            # replicas must stay BIT-identical (the AdamW state has no re-sync): every rank's result equals rank 0's
            both = [torch.empty_like(grad) for _ in range(world)]
            dist.all_gather(both, grad)
            ok = ok and all(torch.equal(both[0], b) for b in both[1:])
        out.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sparse_embedding_row_exchange(world):
    """SparseRowReducer (the token-embedding gradient as touched rows: host-side plan over gloo, one all-gather, index_add)
    equals the dense all-reduce, for unequal per-rank row counts, shared rows and changing ids - and is bitwise equal on
    every rank (3 ranks: fp32 addition does not associate, so the summation order must not depend on the rank)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _zero1_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spn4cir_amd.distributed import GradBucketReducer

        class TorchKernels:
            @staticmethod
            def to_bf16(src, dst):
                dst.copy_(src.to(torch.bfloat16))

            @staticmethod
            def sum_ranks(chunks, G, out_):
                out_.copy_(chunks.view(G, -1).float().sum(0).to(torch.bfloat16))

            @staticmethod
            def to_f32(src, dst):
                dst.copy_(src.float())

            @staticmethod
            def sum_ranks_f32(chunks, G, out_):
                acc = torch.zeros_like(out_)
                for r in range(G):
                    acc += chunks.view(G, -1)[r]
                out_.copy_(acc)
        n = 24 * world * 5 + 7                     # the last bucket (7 + a few elements) cannot be split into aligned chunks
        g = torch.Generator().manual_seed(3)
        grads_all = torch.randn(world, n, generator=g)
        p0 = torch.randn(n, generator=g)
        res = {}
        for dtype in ("fp32", "bf16"):
            params, flat = p0.clone(), grads_all[rank].clone()
            m = torch.zeros(n)
            touched = []

            def update(lo, hi, grad, params=params, m=m, touched=touched):
                # a stateful elementwise rule (momentum + decay): only the owner's slice of the state is ever touched
                m[lo:hi].mul_(0.9).add_(grad, alpha=0.1)
                params[lo:hi].mul_(1 - 0.01).sub_(m[lo:hi], alpha=0.5)
                touched.append((lo, hi))
            red = GradBucketReducer(flat, None, bucket_elems=24 * world, comm_dtype=dtype, kernels=TorchKernels,
                                    shard_update=update, flat_params=params)
            cuts = [n, 24 * world * 5, 24 * world * 3, 24 * world, 0]                  # spans arrive from the end, as in backward
            for k, (hi, lo) in enumerate(zip(cuts[:-1], cuts[1:])):
                red.on_span_ready(lo, hi)
                if k == 0:
                    red.flush()                                                        # the ragged tail is a bucket of its own
            inflight = red.finish_unsharded()
            rest = red.complement_spans(n)
            for lo, hi in rest:                                                        # fallback bucket: all-reduced, updated everywhere
                update(lo, hi, flat[lo:hi])
            for w in inflight:
                w.wait()
            # single-process reference: rank-order sums (fp32), or bf16(sum of bf16-rounded contributions) for the bf16 exchange
            gsum = torch.zeros(n)
            if dtype == "bf16":
                gsum = grads_all.to(torch.bfloat16).float().sum(0).to(torch.bfloat16).float()
            else:
                for r in range(world):
                    gsum += grads_all[r]
            want = p0 * (1 - 0.01) - 0.5 * (0.1 * gsum)
            owned = sum(hi - lo for lo, hi in touched)
            sharded_elems = 24 * world * 5
            ok_bits = torch.equal(params[:sharded_elems], want[:sharded_elems])
            ok_tail = rest == [(sharded_elems, n)] and torch.allclose(params[sharded_elems:], (p0 * 0.99 - 0.05 * grads_all.sum(0))[sharded_elems:],
                                                                      atol=1e-6)
            res[dtype] = (ok_bits, ok_tail, owned == sharded_elems // world + (n - sharded_elems), params.numpy().copy())   # by value: the worker exits right after
        out.put((rank, res["fp32"][:3], res["bf16"][:3], res["fp32"][3], res["bf16"][3]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimizer_update(world):
    """ZeRO-1 shape of the bucket exchange (GradBucketReducer(shard_update=...), Stage2Trainer(optim="sharded")): all-to-all ->
    rank-order sum -> the OWNER updates its 1 / G chunk -> all-gather of the updated masters.  Every rank ends with the same bits,
    equal to a single process applying the rule to the rank-order gradient sum; only 1 / G of the parameters (plus the fallback
    bucket) are updated per rank.  clip4cir/train_negplus.py:77-84,121-123 is the single-device loop this distributes."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero1_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, f32, b16, _, _ in res:
        assert all(f32), ("fp32", rank, f32)
        assert all(b16), ("bf16", rank, b16)
    for r in res[1:]:                                   # replicas bit-identical, both exchange flavours
        assert (r[3] == res[0][3]).all() and (r[4] == res[0][4]).all()
