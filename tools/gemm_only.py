import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
bf = torch.bfloat16
if kind == "nt":
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
    for _ in range(5): ops.gemm_nt(a, b)
else:
    a = torch.randn(M, N, device="cuda").to(bf); b = torch.randn(M, K, device="cuda").to(bf)
    for _ in range(5): ops.gemm_tn(a, b)
torch.cuda.synchronize()
