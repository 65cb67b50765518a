"""bench.py's N-rank control flow on a 1-GPU box: two ranks share device 0 over gloo (SPN_BENCH_SHARE_GPU=1), so that the
sharding of the triplets, the bank-mode selection and its alternative measurement, the touched-row embedding exchange, the
max-over-ranks timing and the single JSON line of rank 0 run end to end (the printed rate is meaningless)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "sharded"])
def test_bench_two_ranks_shared_gpu(mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, SPN_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-recall", "--no-cpu-baseline", "--no-packed", "--bank-mode", mode]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["global_batch"] == 512
    assert d["metric"] == "triplets/sec" and d["value"] > 0 and d["scaling"] == "weak"
    assert "bank_mode_alt" in d and "roofline" in d
    # strong scaling (B_global = 256 split over the ranks) and the bf16 gradient exchange, measured with the same protocol
    st = d["strong"]
    assert st["scaling"] == "strong" and st["global_batch"] == 256 and st["batch_per_gpu"] == 128 and st["value"] > 0
    assert abs(st["value"] - 256 / (st["ms_per_step"] * 1e-3)) < 1e-2 * st["value"]
    assert d["grad_comm_bf16"]["grad_comm_dtype"] == "bf16" and d["grad_comm_bf16"]["value"] > 0
    assert d["grad_comm_direct_fp32"]["grad_comm_algo"] == "direct" and d["grad_comm_direct_fp32"]["value"] > 0
    # sharded optimizer step (ZeRO-1 shape), weak and strong shapes, beside the replicated update
    assert d["optim_sharded"]["optim"] == "sharded" and d["optim_sharded"]["value"] > 0, d["optim_sharded"]
    assert d["optim_sharded_strong"]["optim"] == "sharded" and d["optim_sharded_strong"]["batch_per_gpu"] == 128
    # achieved all-reduce bandwidth and overlap of the gradient exchange with backward (meaningless on a shared GPU: shape only)
    c = d["collectives"]
    assert "error" not in c, c
    assert c["exchange_alone_ms"] > 0 and c["allreduce_busbw_GBps"] > 0 and c["allgather_queries_us"] > 0
    assert c["step_without_exchange_ms"] > 0 and 0.0 <= c["overlap_fraction"] <= 1.0 and c["dense_gradient_bytes"] > 0
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher in front (how a driver may call it): the parent starts
    torch.distributed.run as a child process before touching the GPU, relays rank 0's JSON line last and returns rc 0."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SPN_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-recall",
           "--no-cpu-baseline", "--no-packed", "--no-alt-bank-mode", "--no-extra-configs"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 512 and d["value"] > 0
