"""Per-kernel timing on the GPU box (dev tool): config-2 shapes (T = 256*77 = 19712, W = 768)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def main():
    T, W = 19712, 768
    dev = "cuda"
    bf = torch.bfloat16
    print("== gemm_nt (M,N,K)")
    for (M, N, K) in [(T, 3 * W, W), (T, W, W), (T, 4 * W, W), (T, W, 4 * W), (T, W, 3 * W), (4096, 4096, 4096), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=dev).to(bf); b = torch.randn(N, K, device=dev).to(bf)
        t = timeit(lambda: ops.gemm_nt(a, b))
        print(f"  NT {M}x{N}x{K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF")
    bias = torch.randn(4 * W, device=dev)
    a = torch.randn(T, W, device=dev).to(bf); b = torch.randn(4 * W, W, device=dev).to(bf)
    t = timeit(lambda: ops.gemm_nt(a, b, bias, act=ops.ACT_QUICKGELU, want_pre=True))
    print(f"  NT qgelu {T}x{4*W}x{W}: {t*1e6:8.1f} us  {2*T*4*W*W/t/1e12:7.1f} TF")
    res = torch.randn(T, W, device=dev); a2 = torch.randn(T, 4 * W, device=dev).to(bf); b2 = torch.randn(W, 4 * W, device=dev).to(bf)
    t = timeit(lambda: ops.gemm_nt_resid(a2, b2, bias[:W].contiguous(), res))
    print(f"  NT resid {T}x{W}x{4*W}: {t*1e6:8.1f} us  {2*T*4*W*W/t/1e12:7.1f} TF")
    print("== gemm_tn (Kr,N1,N2)")
    for (Kr, N1, N2) in [(T, W, 4 * W), (T, 4 * W, W), (T, W, W), (T, 3 * W, W)]:
        a = torch.randn(Kr, N1, device=dev).to(bf); b = torch.randn(Kr, N2, device=dev).to(bf)
        t = timeit(lambda: ops.gemm_tn(a, b))
        print(f"  TN {Kr}x{N1}x{N2}: {t*1e6:8.1f} us  {2*Kr*N1*N2/t/1e12:7.1f} TF")
    print("== layernorm")
    x = torch.randn(T, W, device=dev); g = torch.randn(W, device=dev); be = torch.randn(W, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, g, be))
    print(f"  fwd: {t*1e6:8.1f} us  {(T*W*6)/t/1e9:7.1f} GB/s")
    y, mean, rstd = ops.layernorm_fwd(x, g, be)
    dy = torch.randn(T, W, device=dev).to(bf); acc = torch.zeros(T, W, device=dev)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx_accum=acc))
    print(f"  bwd: {t*1e6:8.1f} us  {(T*W*(2+4+4+4+2))/t/1e9:7.1f} GB/s")
    print("== attention (B=256,H=12,L=77 causal)")
    qkv = torch.randn(T, 3 * W, device=dev).to(bf)
    q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
    t = timeit(lambda: ops.attention_fwd(q, k, v, 256, 12, 77, 77, causal=True))
    print(f"  fwd: {t*1e6:8.1f} us")
    o, lse = ops.attention_fwd(q, k, v, 256, 12, 77, 77, causal=True)
    do = torch.randn(T, W, device=dev).to(bf)
    t = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, 256, 12, 77, 77, causal=True))
    print(f"  bwd: {t*1e6:8.1f} us")
    print("== colsum / cast")
    xb = torch.randn(T, 4 * W, device=dev).to(bf)
    t = timeit(lambda: ops.colsum(xb)); print(f"  colsum {T}x{4*W}: {t*1e6:8.1f} us  {T*4*W*2/t/1e9:7.1f} GB/s")
    print("== bank (M=40000, D=768)")
    M, D = 40000, 768
    bank_f = torch.nn.functional.normalize(torch.randn(M, D, device=dev))
    bank = ops.prepare_bank(bank_f)
    bank8 = ops.prepare_bank(bank_f, dtype="fp8")
    for B in (32, 256):
        text = torch.randn(B, D, device=dev)
        q32, qb, inv = ops.combine_l2norm_fwd(None, None, text)
        labels = torch.randint(0, M, (B,), device=dev)
        t = timeit(lambda: ops.bank_stats_fwd(qb, bank, labels, 50.0))
        print(f"  B={B} stats fwd: {t*1e6:8.1f} us  {M*D*2/t/1e9:7.1f} GB/s (bank bytes)  {2*B*M*D/t/1e12:6.1f} TF")
        st = ops.bank_stats_fwd(qb, bank, labels, 50.0)
        lse, row, mean = ops.bank_loss_finalize(st, M)
        t = timeit(lambda: ops.bank_grad_q(qb, bank, labels, 50.0, lse, 1.0 / B))
        print(f"  B={B} grad_q:    {t*1e6:8.1f} us  {M*D*2/t/1e9:7.1f} GB/s (bank bytes)  {4*B*M*D/t/1e12:6.1f} TF")
        t = timeit(lambda: ops.bank_stats_fwd(qb, bank8, labels, 50.0))
        print(f"  B={B} stats fwd fp8 bank: {t*1e6:8.1f} us  {M*D/t/1e9:7.1f} GB/s (bank bytes)")
        t = timeit(lambda: ops.bank_grad_q(qb, bank8, labels, 50.0, lse, 1.0 / B))
        print(f"  B={B} grad_q    fp8 bank: {t*1e6:8.1f} us  {M*D/t/1e9:7.1f} GB/s (bank bytes)")
    print("== adamw 123.65M")
    n = 123650304
    p = torch.randn(n, device=dev); gr = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); vv = torch.zeros(n, device=dev)
    t = timeit(lambda: ops.adamw_step(p, gr, m, vv, 1, 2e-5))
    print(f"  adamw: {t*1e6:8.1f} us  {n*28/t/1e9:7.1f} GB/s")


if __name__ == "__main__":
    main()
