"""Back-to-back GEMMs for a few seconds while rocm-smi is sampled from a child process: is the matrix pipe power-limited?
    python tools/power_gemm.py M N K [seconds]        (SPN_GEMM_CFG etc. select the kernel)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
bf = torch.bfloat16
a = (torch.randn(M, K, device="cuda")).to(bf); b = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
for _ in range(20): ops.gemm_nt(a, b)
torch.cuda.synchronize()
mon = subprocess.Popen(["bash", "-c", "sleep 1.0; for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | sed 's/.*: //' | tr '\\n' ' '; echo; sleep 0.5; done"],
                       stdout=subprocess.PIPE, text=True)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < secs:
    for _ in range(200): ops.gemm_nt(a, b)
    torch.cuda.synchronize(); n += 200
dt = time.perf_counter() - t0
print(f"M={M} N={N} K={K} cfg={os.environ.get('SPN_GEMM_CFG','3')}: {dt/n*1e6:.1f} us  {2*M*N*K*n/dt/1e12:.0f} TFLOP/s")
print(mon.communicate()[0])
