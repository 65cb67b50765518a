"""Two ranks sharing the one GPU of the test box (gloo carries the collectives; RCCL refuses two
ranks per device): the full fused step - tower, sharded/replicated bank loss, bucketed gradient
all-reduce, AdamW - must reproduce the single-process step on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

W, LAYERS, D, VOCAB, B, M, TAU, LR = 128, 2, 128, 600, 16, 901, 0.03, 1e-3


def _setup():
    from spn4cir_amd import synthetic
    layers = int(os.environ.get("SPN_TEST_DDP_LAYERS", LAYERS))     # inherited by the spawned ranks
    sd = synthetic.text_state_dict(W, layers, D, vocab=VOCAB, seed=0)
    target, refer = synthetic.banks(M, D)
    ids = synthetic.token_ids(B, vocab=VOCAB, seed=1)
    ridx, labels = synthetic.triplet_indices(B, M)
    return sd, target, refer, ids, ridx, labels


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
        from spn4cir_amd.models import CIRPlus
        from spn4cir_amd.trainer import Stage2Trainer
        sd, target, refer, ids, ridx, labels = _setup()
        dev = torch.device("cuda", 0)
        model = CIRPlus(sd, tau=TAU, device=dev, plus=True)
        comm = os.environ.get("SPN_TEST_DDP_COMM", "fp32")
        tr = Stage2Trainer(model, lr=LR, group=None, bank_mode=mode, grad_comm_dtype="bf16" if comm.endswith("bf16") else "fp32",
                           grad_comm_algo="direct" if comm == "direct" else None,
                           optim="sharded" if comm.startswith("zero1") else "replicated")
        tr.set_banks(refer, target)
        bl = B // world
        sl = slice(rank * bl, (rank + 1) * bl)
        losses = []
        # SPN_TEST_DDP_SPARSE=1: the token-embedding gradient goes out as touched rows (Stage2Trainer.step(ids_host=))
        host = ids[sl].contiguous() if os.environ.get("SPN_TEST_DDP_SPARSE") == "1" else None
        for _ in range(2):
            losses.append(tr.step(ids[sl].to(dev), ridx[sl].to(dev), labels[sl].to(dev), ids_host=host).item())
        out.put((rank, losses, model.tower.params.cpu().numpy()))   # by value: the worker exits right after
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,layers,sparse,comm", [("sharded", 2, 0, "fp32"), ("replicated", 2, 0, "fp32"),
                                                     ("sharded", 6, 0, "fp32"),   # 6 blocks: three weight-gradient groups (4 + 1 + 1) per rank
                                                     ("replicated", 6, 1, "fp32"), ("sharded", 2, 1, "fp32"),   # embedding rows exchanged sparsely
                                                     # gradient buckets as bf16: all-to-all + fp32 sum in rank order + all-gather
                                                     ("replicated", 2, 0, "bf16"), ("replicated", 6, 1, "bf16"),
                                                     # fp32 buckets through the same direct exchange instead of the ring all-reduce
                                                     ("sharded", 6, 0, "direct"),
                                                     # sharded optimizer step (ZeRO-1 shape): the owner of a reduced chunk updates it,
                                                     # the updated masters are all-gathered (Stage2Trainer(optim="sharded"))
                                                     ("replicated", 6, 0, "zero1"), ("sharded", 2, 1, "zero1"),
                                                     ("replicated", 6, 1, "zero1_bf16")])
def test_two_ranks_match_single_process(mode, layers, sparse, comm, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("SPN_TEST_DDP_SPARSE", str(sparse))
    monkeypatch.setenv("SPN_TEST_DDP_COMM", comm)
    monkeypatch.setenv("SPN_TEST_DDP_LAYERS", str(layers))
    from spn4cir_amd.models import CIRPlus
    from spn4cir_amd.trainer import Stage2Trainer
    sd, target, refer, ids, ridx, labels = _setup()
    dev = torch.device("cuda", 0)
    model = CIRPlus(sd, tau=TAU, device=dev, plus=True)
    tr = Stage2Trainer(model, lr=LR)
    tr.set_banks(refer, target)
    ref_losses = [tr.step(ids.to(dev), ridx.to(dev), labels.to(dev)).item() for _ in range(2)]
    ref_params = model.tower.params.cpu()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = [(r, l, torch.from_numpy(p)) for r, l, p in res]
    for rank, losses, params in res:
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) < 2e-3 * max(1.0, abs(b)), (mode, rank, losses, ref_losses)
        # same update as the single-process run: differences only from bf16 rounding of partial sums.  AdamW moves every
        # element by ~lr per step whatever the gradient's size, so an element whose gradient is rounding noise can differ
        # by 2 lr per step (opposite signs); the deeper model has a few of those: bound = 2 steps x 2 lr, and all but a
        # handful of elements within the 2-block bound
        diff = (params - ref_params).abs()
        assert diff.max() < (2e-3 if layers == 2 else 4.2e-3), (mode, rank, diff.max().item())
        assert (diff > 2e-3).float().mean() < 1e-4
    assert (res[0][2] - res[1][2]).abs().max() == 0.0       # replicas stay bit-identical


def _zero1_bits_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spn4cir_amd import ops
        from spn4cir_amd.distributed import GradBucketReducer
        dev = torch.device("cuda", 0)
        n = 3 * 1024 * 1024 + 8 * world                    # three buckets of 1 Mi elements + a short tail bucket
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(n, generator=g).to(dev)
        p0 = torch.randn(n, generator=torch.Generator().manual_seed(7)).to(dev)
        res = {}
        for dtype in ("fp32", "bf16"):
            # (a) replicated: direct exchange of the gradients, AdamW over everything on every rank
            flat, pa = grads.clone(), p0.clone()
            ma, va = torch.zeros_like(pa), torch.zeros_like(pa)
            red = GradBucketReducer(flat, None, bucket_elems=1 << 20, comm_dtype=dtype, algo="direct")
            for lo in range(n - 8 * world, -1, -(1 << 20)):
                red.on_span_ready(lo, min(n, lo + (1 << 20)) if lo < n - 8 * world else n)
            red.finish()
            ops.adamw_step(pa, flat, ma, va, 1, 1e-3)
            # (b) sharded: the owner updates its chunk of every bucket, the masters are all-gathered
            flat2, pb = grads.clone(), p0.clone()
            mb, vb = torch.zeros_like(pb), torch.zeros_like(pb)
            owned = []

            def update(lo, hi, grad, pb=pb, mb=mb, vb=vb, owned=owned):
                ops.adamw_step(pb[lo:hi], grad, mb[lo:hi], vb[lo:hi], 1, 1e-3)
                owned.append(hi - lo)
            red2 = GradBucketReducer(flat2, None, bucket_elems=1 << 20, comm_dtype=dtype, shard_update=update, flat_params=pb)
            for lo in range(n - 8 * world, -1, -(1 << 20)):
                red2.on_span_ready(lo, min(n, lo + (1 << 20)) if lo < n - 8 * world else n)
            inflight = red2.finish_unsharded()
            for lo, hi in red2.complement_spans(n):
                ops.adamw_step(pb[lo:hi], flat2[lo:hi], mb[lo:hi], vb[lo:hi], 1, 1e-3)
            for w in inflight:
                w.wait()
            torch.cuda.synchronize()
            res[dtype] = (int((pa != pb).sum().item()), sum(owned), pb.cpu().numpy())
        out.put((rank, res["fp32"][:2], res["bf16"][:2], res["fp32"][2], res["bf16"][2]))
    finally:
        dist.destroy_process_group()


def test_sharded_optimizer_update_bits_on_the_hip_kernels():
    """optim="sharded" on the library's kernels (two ranks sharing the GPU over gloo): all-to-all -> spn_sum_ranks_* -> spn_adamw_step
    on the OWNED chunk -> all-gather of the masters gives, bit for bit, the parameters of the replicated update (direct exchange of
    the gradients + AdamW over everything), on both ranks, for fp32 and bf16 payloads - and each rank updated 1 / G of the buckets
    (train_negplus.py:77-84,121-123 is the single-device loop this distributes)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_zero1_bits_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n = 3 * 1024 * 1024 + 8 * world
    for rank, f32, b16, _, _ in res:
        assert f32 == (0, n // world), ("fp32", rank, f32)
        assert b16 == (0, n // world), ("bf16", rank, b16)
    assert (res[0][3] != res[1][3]).sum() == 0 and (res[0][4] != res[1][4]).sum() == 0       # replicas bit-identical


_NCCL_WORLD1 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_trainer_ddp_gpu import _setup, TAU, LR
from spn4cir_amd.models import CIRPlus
from spn4cir_amd.trainer import Stage2Trainer
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sd, target, refer, ids, ridx, labels = _setup()
def run(mode, sparse=False):
    model = CIRPlus(sd, tau=TAU, device=dev, plus=True)
    tr = Stage2Trainer(model, lr=LR, group=None, bank_mode=mode, pack=False)    # dense on both sides: the comparison below is to 1e-6
    tr.set_banks(refer, target)
    # sparse: the touched-row exchange of the token-embedding gradient - the extra gloo group next to NCCL, the async
    # all_gather_into_tensor and the index_add_ on the RCCL stream
    ls = [tr.step(ids.to(dev), ridx.to(dev), labels.to(dev), ids_host=ids if sparse else None).item() for _ in range(2)]
    return ls, model.tower.params.clone()
ref_l, ref_p = run("replicated")                      # no process group yet: plain single-GPU step
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mode, sparse in (("sharded", False), ("replicated", False), ("replicated", True)):
    l, p = run(mode, sparse)                          # every collective now goes through RCCL
    assert max(abs(a - b) for a, b in zip(l, ref_l)) < 1e-5, (mode, sparse, l, ref_l)
    assert (p - ref_p).abs().max().item() < 1e-6, (mode, sparse)
# the bf16 bucket exchange on RCCL (all_to_all_single + all_gather_into_tensor on the reducer's own stream, the library's cast /
# rank-sum kernels between them): with one rank the "sum" is the bf16 rounding of the gradient - an AdamW step away at most
def run_bf16():
    model = CIRPlus(sd, tau=TAU, device=dev, plus=True)
    tr = Stage2Trainer(model, lr=LR, group=None, bank_mode="replicated", grad_comm_dtype="bf16")
    tr.set_banks(refer, target)
    ls = [tr.step(ids.to(dev), ridx.to(dev), labels.to(dev)).item() for _ in range(2)]
    return ls, model.tower.params.clone()
def run_direct():
    model = CIRPlus(sd, tau=TAU, device=dev, plus=True)
    tr = Stage2Trainer(model, lr=LR, group=None, bank_mode="replicated", pack=False, grad_comm_algo="direct")
    tr.set_banks(refer, target)
    ls = [tr.step(ids.to(dev), ridx.to(dev), labels.to(dev)).item() for _ in range(2)]
    return ls, model.tower.params.clone()
l, p = run_direct()                                   # one rank: the "sum" is the gradient itself - same step as the ring path
assert max(abs(a - b) for a, b in zip(l, ref_l)) < 1e-5 and (p - ref_p).abs().max().item() < 1e-6
def run_zero1():
    model = CIRPlus(sd, tau=TAU, device=dev, plus=True)
    tr = Stage2Trainer(model, lr=LR, group=None, bank_mode="replicated", pack=False, optim="sharded")
    assert tr.optim == "sharded"
    tr.set_banks(refer, target)
    ls = [tr.step(ids.to(dev), ridx.to(dev), labels.to(dev)).item() for _ in range(2)]
    return ls, model.tower.params.clone()
l, p = run_zero1()                                    # the sharded optimizer step's collectives on RCCL: one rank owns everything
assert max(abs(a - b) for a, b in zip(l, ref_l)) < 1e-5 and (p - ref_p).abs().max().item() < 1e-6
l, p = run_bf16()
assert max(abs(a - b) for a, b in zip(l, ref_l)) < 2e-3, (l, ref_l)
assert (p - ref_p).abs().max().item() < 4.2e-3 and ((p - ref_p).abs() > 2e-3).float().mean().item() < 1e-3
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("NCCL_WORLD1_OK")
"""


def test_rccl_call_pattern_on_one_rank():
    """The N-GPU path's RCCL calls (bf16/int64/fp32 all-gathers, reduce_scatter_tensor, async all-reduce of
    in-place bucket slices, barrier) issued for real on the nccl backend with a 1-rank group
    (SPN_DP_FORCE_COLLECTIVES=1); the step must equal the plain single-GPU step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    import sys
    env = dict(os.environ, SPN_DP_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_WORLD1, str(_free_port())], env=env, capture_output=True, text=True,
                       timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "NCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


class _SlicedStub:
    """encode_text stub whose pre-made features follow the rank's query slice (the metric code needs the protocol only)."""

    def __init__(self, feats, begin):
        self.feats, self.pos, self.output_dim = feats, begin, feats.shape[1]

    def encode_text(self, captions):
        out = self.feats[self.pos:self.pos + len(captions)]
        self.pos += len(captions)
        return out


def _val_worker(rank, world, port, golden_dir, out):
    import json
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spn4cir_amd import validate
        from spn4cir_amd.distributed import shard_range
        z = np.load(os.path.join(golden_dir, "recall.npz"))
        names, members = json.loads(str(z["names"])), json.loads(str(z["members"]))
        gallery, text = torch.from_numpy(z["gallery"]).cuda(), torch.from_numpy(z["text_feats"]).cuda()
        ref_idx, tgt_idx = z["ref_idx"], z["tgt_idx"]
        fiq = [(names[r], names[t], [f"a {i}.", f"b {i}?"]) for i, (r, t) in enumerate(zip(ref_idx, tgt_idx))]
        cirr = [(names[r], names[t], f"cap {i}", members[i]) for i, (r, t) in enumerate(zip(ref_idx, tgt_idx))]
        b, _ = shard_range(len(fiq), world, rank)
        r = validate.compute_fiq_val_metrics(fiq, _SlicedStub(text, b), gallery, names, distributed=True)
        c = validate.compute_cirr_val_metrics(cirr, _SlicedStub(text, b), gallery, names, distributed=True)
        out.put((rank, tuple(r), tuple(c)))
    finally:
        dist.destroy_process_group()


def test_query_sharded_validation_matches_reference(golden_dir):
    """validation shards the query set over the ranks (gallery replicated) and sums hit counts - every rank
    must report the reference's FashionIQ / CIRR metrics."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    z = np.load(os.path.join(golden_dir, "recall.npz"))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_val_worker, args=(r, 2, port, golden_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, fiq, cirr in res:
        assert fiq == pytest.approx(tuple(z["fiq"]), abs=1e-9)
        assert cirr == pytest.approx(tuple(z["cirr"]), abs=1e-4)
