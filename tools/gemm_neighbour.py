"""One NT GEMM launch at a time, operands fixed or rotated through 1.2 GB, with / without an HBM-bound kernel in front of every
launch (profiles/r02_nt2_vs_nt3_in_step_per_shape.txt):  [SPN_GEMM_CFG=7] python tools/gemm_neighbour.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spn4cir_amd import ops
M, N, K = 19712, 768, 3072
bf = torch.bfloat16
As = [torch.randn(M, K, device="cuda").to(bf) for _ in range(10)]     # 1.2 GB: beyond the 256 MB infinity cache
b = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
filler = torch.randn(64 << 20, device="cuda")
def bench(rot, with_filler):
    for i in range(10): ops.gemm_nt(As[i % 10 if rot else 0], b)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(60):
        if with_filler: filler.mul_(1.0)            # an HBM-bound neighbour (256 MB read + write) like LayerNorm in the step
        e0.record(); ops.gemm_nt(As[i % 10 if rot else 0], b); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / 60 * 1e3
for rot in (0, 1):
    for fil in (0, 1):
        t = bench(rot, fil)
        print(f"cfg={os.environ.get('SPN_GEMM_CFG','3')} rotate={rot} neighbour={fil}: {t:.1f} us  {2*M*N*K/t/1e6:.0f} TF")
