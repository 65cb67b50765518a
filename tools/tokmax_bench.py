"""Token-max bank loss timing: B x (M targets x 32 tokens x 256) bf16 bank, forward + backward.
Algorithmic HBM bytes per pass = M*32*256*2 (the bank once)."""
import sys, time, torch
sys.path.insert(0, ".")
from spn4cir_amd import ops

def main():
    D = 256
    for B, M in [(32, 20000), (128, 20000), (128, 40000)]:
        g = torch.Generator(device="cuda").manual_seed(0)
        bank = torch.nn.functional.normalize(torch.randn(M, 32, D, device="cuda", generator=g), dim=-1).to(torch.bfloat16)
        q = torch.nn.functional.normalize(torch.randn(B, D, device="cuda", generator=g), dim=-1).to(torch.bfloat16)
        labels = torch.randint(0, M, (B,), device="cuda")
        def fwd():
            st = ops.bank_stats_fwd_tokmax(q, bank, labels, 1 / 0.07)
            return ops.bank_loss_finalize(st, M)
        lse, _, _ = fwd()
        def bwd():
            return ops.bank_grad_q_tokmax(q, bank, labels, 1 / 0.07, lse, 1.0 / B)
        for name, f in (("fwd", fwd), ("bwd", bwd)):
            for _ in range(3): f()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): f()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
            print(f"B={B} M={M} {name}: {dt*1e6:.0f} us  {M*32*D*2/dt/1e12:.2f} TB/s algorithmic")

main()
