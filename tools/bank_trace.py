"""One bank InfoNCE case under rocprofv3 (tools/bank_trace.sh): the forward/backward pair of the default routing, 24
launches rotating over 4 bank copies.    python3 tools/bank_trace.py B M D bf16|fp8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops
B, M, D, dt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
g = torch.Generator().manual_seed(0)
banks = [ops.prepare_bank(torch.nn.functional.normalize(torch.randn(M, D, generator=g)).cuda(), dt) for _ in range(4)]
q = torch.nn.functional.normalize(torch.randn(B, D, generator=g)).cuda()
_, qb, _ = ops.combine_l2norm_fwd(None, None, q)
labels = torch.randint(0, M, (B,), generator=g).cuda()
save = ops.bank_logits_buffer(B, M, "cuda")
for i in range(28):
    stats = ops.bank_stats_fwd(qb, banks[i % 4], labels, 50.0, save=save)
    lse, _, _ = ops.bank_loss_finalize(stats, M)
    ops.bank_grad_q(qb, banks[i % 4], labels, 50.0, lse, 1.0 / B, saved=save)
torch.cuda.synchronize()
