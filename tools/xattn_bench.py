"""Absorbed cross-attention (spn_xattn_fwd / spn_xattn_bwd) at BLIP shapes, per-kernel times under rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o kt -- python3 tools/xattn_bench.py B L [B L ...]
Each (B, L) pair runs 10 forward + backward passes (dense rows, H = 12, S = 577, E = 768); a marker kernel (zero fill of B*L
floats... the per-pair kernel names are the same, so run one pair per process for clean tables)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import ops
bf = torch.bfloat16
H, S, E = 12, 577, 768
W = H * 64
args = [int(a) for a in sys.argv[1:]] or [128, 32]
for B, L in zip(args[0::2], args[1::2]):
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B * L, W, device="cuda", generator=g).to(bf)
    wkv = (torch.randn(2 * W, E, device="cuda", generator=g) * 0.02).to(bf)
    bkv = torch.zeros(2 * W, device="cuda")
    x = torch.randn(B, S, E, device="cuda", generator=g).to(bf)
    dctx = torch.randn(B * L, W, device="cuda", generator=g).to(bf)
    wkv_t = wkv.t().contiguous()
    for _ in range(10):
        ctx, saved = ops.xattn_fwd(q, wkv, bkv, x, H, wkv_t=wkv_t)
        ops.xattn_bwd(saved, dctx)
    torch.cuda.synchronize()
    print("done", B, L)
