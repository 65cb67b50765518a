"""NT GEMMs with few rows - the packed text tower (5 229 live rows -> 5 248), 32 triplets per GPU (2 464 rows), the BERT side of the
BLIP step (4 096 rows) - in the text-tower / BERT shapes and epilogues, kernel time by HIP events (rotating operand copies).
    SPN_NT_MID=<variant> python tools/small_gemm_bench.py      (the routing switch is read when the library loads)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import ops
bf = torch.bfloat16
rows = [2464, 4096, 5248]
shapes = [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc+gelu"), (768, 3072, "proj")]
tot = {}
print("SPN_NT_MID =", os.environ.get("SPN_NT_MID", "0"))
for M in rows:
    for N, K, tag in shapes:
        As = [torch.randn(M, K, device="cuda").to(bf) for _ in range(4)]
        w = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        bias = torch.zeros(N, device="cuda")
        def run(n):
            for i in range(n):
                if tag == "fc+gelu":
                    ops.gemm_nt(As[i % 4], w, bias, act=ops.ACT_QUICKGELU, want_pre=True)
                else:
                    ops.gemm_nt(As[i % 4], w, bias)
        run(5)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); run(50); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        tot[M] = tot.get(M, 0.0) + us
        print(f"  M={M:5d} N={N:4d} K={K:4d} {tag:8s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s")
for M in rows:
    print(f"  M={M}: sum of the four {tot[M]:.1f} us")
