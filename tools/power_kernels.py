"""Socket power and clock while ONE kernel class runs back to back (rocm-smi sampled from a child process).
    python tools/power_kernels.py [tn|ln|attn|adamw|nt]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "tn"
T, W = 19712, 768
dev = "cuda"; bf = torch.bfloat16
if kind == "tn":
    layer = [(768, 3072), (3072, 768), (2304, 768), (768, 768)]
    pairs = [(torch.randn(T, a, device=dev).to(bf), torch.randn(T, b, device=dev).to(bf)) for a, b in layer * 12]
    f = lambda: ops.gemm_tn_grouped(pairs); work = sum(2.0 * T * a.shape[1] * b.shape[1] for a, b in pairs); unit = "TFLOP/s"
elif kind == "nt":
    a = torch.randn(T, 768, device=dev).to(bf); b = torch.randn(2304, 768, device=dev).to(bf)
    f = lambda: ops.gemm_nt(a, b); work = 2.0 * T * 2304 * 768; unit = "TFLOP/s"
elif kind == "ln":
    x = torch.randn(T, W, device=dev); dy = torch.randn(T, W, device=dev).to(bf); g = torch.randn(W, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, g)
    dx = torch.zeros(T, W, device=dev)
    f = lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx_accum=dx); work = T * W * 16.0 / 1e0; unit = "TB/s"
elif kind == "attn":
    qkv = torch.randn(T, 3 * W, device=dev).to(bf); do = torch.randn(T, W, device=dev).to(bf)
    q, k, v = qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:]
    o, lse = ops.attention_fwd(q, k, v, 256, 12, 77, 77, causal=True)
    f = lambda: ops.attention_bwd(q, k, v, o, lse, do, 256, 12, 77, 77, causal=True); work = T * W * 14.0; unit = "TB/s"
else:
    n = 123_650_305
    p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3; m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    f = lambda: ops.adamw_step(p, g, m, v, 1, 2e-5); work = n * 28.0; unit = "TB/s"
for _ in range(5): f()
torch.cuda.synchronize()
mon = subprocess.Popen(["bash", "-c", "sleep 1.0; for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | sed 's/.*: //' | tr '\\n' ' '; echo; sleep 0.5; done"],
                       stdout=subprocess.PIPE, text=True)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 3.5:
    for _ in range(20): f()
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
print(f"{kind}: {dt / n * 1e6:.1f} us per call  {work * n / dt / 1e12:.2f} {unit}")
print(mon.communicate()[0])
