import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import ops
B, M, D = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 40000, 768)))
g = torch.Generator().manual_seed(0)
bank = ops.prepare_bank(torch.nn.functional.normalize(torch.randn(M, D, generator=g)).cuda())
q = torch.nn.functional.normalize(torch.randn(B, D, generator=g)).cuda()
_, qb, _ = ops.combine_l2norm_fwd(None, None, q)
labels = torch.randint(0, M, (B,), generator=g).cuda()
save = ops.bank_logits_buffer(B, M, "cuda")
for i in range(6):
    stats = ops.bank_stats_fwd(qb, bank, labels, 50.0, save=save)
    lse, _, _ = ops.bank_loss_finalize(stats, M)
    ops.bank_grad_q(qb, bank, labels, 50.0, lse, 1.0 / B, saved=save)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for i in range(20):
    ev[0].record()
    stats = ops.bank_stats_fwd(qb, bank, labels, 50.0, save=save)
    lse, _, _ = ops.bank_loss_finalize(stats, M)
    ev[1].record()
    ops.bank_grad_q(qb, bank, labels, 50.0, lse, 1.0 / B, saved=save)
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print("B", B, "M", M, "D", D, "fwd+finalize / bwd us (event pairs, launch gaps included)", round(tf * 50, 1), round(tb * 50, 1))
