"""Reference point only (not used by the product): what the vendor GEMM reaches on the tower shapes."""
import torch
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
T, W = 19712, 768
for (M, N, K) in [(T, 3*W, W), (T, W, W), (T, 4*W, W), (T, W, 4*W), (T, W, 3*W), (4096, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: torch.matmul(a, b.t()))
    print(f"  torch NT {M}x{N}x{K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF")
for (Kr, N1, N2) in [(T, W, 4*W), (T, 4*W, W), (T, W, W), (T, 3*W, W)]:
    a = torch.randn(Kr, N1, device="cuda", dtype=torch.bfloat16); b = torch.randn(Kr, N2, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: torch.matmul(a.t(), b))
    print(f"  torch TN {Kr}x{N1}x{N2}: {t*1e6:8.1f} us  {2*Kr*N1*N2/t/1e12:7.1f} TF")
