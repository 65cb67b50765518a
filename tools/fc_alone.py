"""The fc + QuickGELU product of the text tower alone (19 712 x 3 072 x 768, two bf16 outputs), back to back:
   python3 tools/fc_alone.py [n_rotating_A_buffers]   (1 = the same operand every launch; 16 = 480 MB of A operands, past the MALL)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import ops
nrot = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M, N, K = 19712, 3072, 768
bf = torch.bfloat16
As = [torch.randn(M, K, device="cuda").to(bf) for _ in range(nrot)]
w = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
bias = torch.zeros(N, device="cuda")
def run(n):
    for i in range(n):
        ops.gemm_nt(As[i % nrot], w, bias, act=ops.ACT_QUICKGELU, want_pre=True)
run(5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record(); run(40); ev[1].record(); torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1000 / 40
print("fc + QuickGELU alone, %d rotating A: %.1f us per launch, %.0f TFLOP/s" % (nrot, us, 2.0 * M * N * K / us / 1e6))
