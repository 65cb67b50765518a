"""Times the whole-head attention kernels at the text tower's shape (B sequences x H heads x L tokens, packed qkv rows).
   python3 tools/attn_bench.py [B H L]        SPN_LIB_PATH selects an ablation build of the library."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import ops
B, H, L = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 12, 77)))
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * L, 3 * H * 64, generator=g) * 0.5).to(torch.bfloat16).cuda()
q, k, v = qkv[:, :H * 64], qkv[:, H * 64:2 * H * 64], qkv[:, 2 * H * 64:]
d_o = (torch.randn(B * L, H * 64, generator=g) * 0.1).to(torch.bfloat16).cuda()
o, lse = ops.attention_fwd(q, k, v, B, H, L, L, causal=True)
def timed(f, n=30):
    for _ in range(5):
        f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        f()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1000 / n
tf = timed(lambda: ops.attention_fwd(q, k, v, B, H, L, L, causal=True))
tb = timed(lambda: ops.attention_bwd(q, k, v, o, lse, d_o, B, H, L, L, causal=True))
print(os.environ.get("SPN_LIB_PATH", "shipped").split("/")[-1], "B H L", B, H, L, "fwd us", round(tf, 1), "bwd us", round(tb, 1))
