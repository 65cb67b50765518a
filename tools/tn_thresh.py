"""gemm_tn: v1 (128x128) vs v2 (256x256 staggered) on the weight-gradient shapes of W = 512 / 640 towers."""
import sys, time, torch
sys.path.insert(0, ".")
from spn4cir_amd import ops
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
T = 19712
for (N1, N2) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (1920, 640), (2560, 640), (640, 2560), (768, 768), (2304, 768)]:
    a = torch.randn(T, N1, device="cuda").to(torch.bfloat16); b = torch.randn(T, N2, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: ops.gemm_tn(a, b))
    print(f"TN {T}x{N1}x{N2}: {t*1e6:7.1f} us {2*T*N1*N2/t/1e12:7.1f} TF")
