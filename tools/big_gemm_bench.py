"""The four NT products of a text-tower block at config 2's 19 712 rows (+ fc's QuickGELU epilogue), kernel time by HIP events.
    [SPN_LIB_PATH=... SPN_NT_MID=v SPN_NT_MID_ALL=1] python tools/big_gemm_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spn4cir_amd import ops
bf = torch.bfloat16
M = 19712
print("SPN_NT_MID =", os.environ.get("SPN_NT_MID", "0"), "ALL =", os.environ.get("SPN_NT_MID_ALL", "0"))
tot = 0.0
for N, K, tag in [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc+gelu"), (768, 3072, "proj")]:
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
    e0.record(); run(40); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 40
    tot += us
    print(f"  M={M} N={N:4d} K={K:4d} {tag:8s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s")
print(f"  sum {tot:.1f} us")
