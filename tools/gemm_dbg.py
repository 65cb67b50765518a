"""Bottleneck elimination on the NT GEMM: time shapes under SPN_GEMM_DBG bit masks (results are wrong by design).
Needs a probe build: tools/build_variant.sh probes "-DSPN_GEMM_PROBES -DSPN_GEMM_LOOP_DBG=1" and
SPN_LIB_PATH=spn4cir_amd/libspn4cir_hip_probes.so (the shipped library compiles the probes out)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(19712, 3072, 768), (19712, 768, 3072), (19712, 768, 768), (8192, 8192, 8192)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from spn4cir_amd import ops
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
        for _ in range(3): ops.gemm_nt(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm_nt(a, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        extra = ""
        if int(os.environ.get("SPN_GEMM_DBG", "0")) & 32:
            out = ops.gemm_nt(a, b); torch.cuda.synchronize()
            c = out.view(-1)[:4].view(torch.int32).cpu().tolist()
            cyc, ticks = c[0] & 0xffffffff, c[1] & 0xffffffff
            extra = f"  main loop of the last tile: {cyc} shader cycles in {ticks * 10} ns = {cyc / max(ticks, 1) * 100:.0f} MHz"
        print(f"  {M}x{N}x{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF{extra}", flush=True)
else:
    for dbg in (sys.argv[1:] or ["0", "1", "2", "4", "6", "8", "14"]):
        print(f"== SPN_GEMM_DBG={dbg} CFG={os.environ.get('SPN_GEMM_CFG', 'default')}", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SPN_GEMM_DBG=dbg))
