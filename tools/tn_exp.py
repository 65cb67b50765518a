"""Time the weight-gradient (TN) GEMMs of one layer in isolation: tools/tn_exp.py (use SPN_LIB_PATH for experiment builds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spn4cir_amd import ops
T = 19712
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N1, N2 in ((768, 3072), (3072, 768), (2304, 768), (768, 768)):
    a = torch.randn(T, N1, device="cuda").bfloat16(); b = torch.randn(T, N2, device="cuda").bfloat16()
    us = timeit(lambda: ops.gemm_tn(a, b, want_colsum=True))
    print(f"  TN {T} x {N1} x {N2}: {us:7.1f} us  {2.0 * T * N1 * N2 / us / 1e6:7.1f} TF (incl. split-K reduce)", flush=True)
a1 = torch.randn(T, 2304, device="cuda").bfloat16(); b1 = torch.randn(T, 768, device="cuda").bfloat16()
a2 = torch.randn(T, 768, device="cuda").bfloat16(); b2 = torch.randn(T, 768, device="cuda").bfloat16()
us = timeit(lambda: ops.gemm_tn_pair(a1, b1, a2, b2))
print(f"  TN pair 2304x768 + 768x768: {us:7.1f} us  {2.0 * T * (2304 * 768 + 768 * 768) / us / 1e6:7.1f} TF", flush=True)
# no-split emulation: one wide problem with >= 256 tiles (what a grouped launch over several layers would look like)
for N1, N2 in ((9216, 3072), (5376, 3072), (3072, 3072 * 3)):
    a = torch.randn(T, N1, device="cuda").bfloat16(); b = torch.randn(T, N2, device="cuda").bfloat16()
    us = timeit(lambda: ops.gemm_tn(a, b, want_colsum=True), n=5)
    tiles = (N1 // 256) * (N2 // 256)
    print(f"  TN {T} x {N1} x {N2} ({tiles} tiles, {tiles / 256:.2f} rounds): {us:8.1f} us  {2.0 * T * N1 * N2 / us / 1e6:7.1f} TF", flush=True)
# grouped launch over L layers' worth of weight gradients (no split-K) against the per-problem launches
layer = [(768, 3072), (3072, 768), (2304, 768), (768, 768)]
for L in (1, 2, 5, 7, 12):
    pairs = [(torch.randn(T, n1, device="cuda").bfloat16(), torch.randn(T, n2, device="cuda").bfloat16()) for n1, n2 in layer * L]
    us = timeit(lambda: ops.gemm_tn_grouped(pairs), n=3)
    fl = sum(2.0 * T * a.shape[1] * b.shape[1] for a, b in pairs)
    tiles = sum((a.shape[1] // 256) * (b.shape[1] // 256) for a, b in pairs)
    print(f"  grouped TN, {L:2d} layers ({tiles} tiles): {us:9.1f} us = {us / L:7.1f} us per layer  {fl / us / 1e6:7.1f} TF", flush=True)
    del pairs
