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
elif kind == "tng":     # the grouped weight-gradient launch of the step: 12 blocks x 4 products over M token rows
    layer = [(768, 3072), (3072, 768), (2304, 768), (768, 768)]
    pairs = [(torch.randn(M, n1, device="cuda").to(bf), torch.randn(M, n2, device="cuda").to(bf)) for n1, n2 in layer * 12]
    for _ in range(3): ops.gemm_tn_grouped(pairs)
else:
    a = torch.randn(M, N, device="cuda").to(bf); b = torch.randn(M, K, device="cuda").to(bf)
    for _ in range(5): ops.gemm_tn(a, b)
torch.cuda.synchronize()
