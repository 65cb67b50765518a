"""Correctness + speed of the hand-scheduled 4-wave NT GEMM (SPN_GEMM_CFG=7) against the default kernel (child
processes: the configuration is read once per process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(256, 256, 768), (19712, 3072, 768), (19712, 768, 3072), (19712, 768, 768), (19712, 2304, 768), (19712, 768, 2304), (8192, 8192, 8192)]
CHECK = [(256, 256, 64), (256, 512, 128), (512, 256, 192), (512, 768, 256), (1000, 520, 320), (19712, 768, 768)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from spn4cir_amd import ops
    torch.manual_seed(0)
    if sys.argv[2] == "check":
        def qg(x): return x * torch.sigmoid(1.702 * x)
        def report(name, out, ref, tol):
            err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
            bad = (out.float() - ref).abs() > tol * ref.abs().max()
            print(f"  check {name}: max rel err {err:.2e}  bad elements {int(bad.sum())}", flush=True)
            if bad.any():
                idx = bad.nonzero()
                print("   first bad:", idx[:5].tolist(), " rows bad:", idx[:, 0].unique().numel(), "cols bad:", idx[:, 1].unique().numel())
        for M, N, K in CHECK:
            a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
            bias = torch.randn(N, device="cuda")
            ref = a.float() @ b.float().t()
            for rep in range(2):
                report(f"{M}x{N}x{K} bf16 rep {rep}", ops.gemm_nt(a, b), ref, 1e-2)
                report(f"{M}x{N}x{K} bf16+bias rep {rep}", ops.gemm_nt(a, b, bias=bias), ref + bias, 1e-2)
                report(f"{M}x{N}x{K} f32+bias rep {rep}", ops.gemm_nt(a, b, bias=bias, out_dtype=torch.float32), ref + bias, 2e-3)
                u, pre = ops.gemm_nt(a, b, bias=bias, act=ops.ACT_QUICKGELU, want_pre=True)
                report(f"{M}x{N}x{K} gelu pre rep {rep}", pre, ref + bias, 1e-2)
                report(f"{M}x{N}x{K} gelu out rep {rep}", u, qg(ref + bias), 1e-2)
                resid = torch.randn(M, N, device="cuda")
                report(f"{M}x{N}x{K} resid rep {rep}", ops.gemm_nt_resid(a, b, bias, resid), ref + bias + resid, 2e-3)
                p = torch.randn(M, N, device="cuda").bfloat16()
                x = p.float().requires_grad_(True); qg(x).sum().backward()
                report(f"{M}x{N}x{K} dact rep {rep}", ops.gemm_nt_dact(a, b, p, ops.ACT_QUICKGELU), ref * x.grad, 1e-2)
    else:
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
                extra = (f"  k loop of the last tile: {cyc} cycles = {cyc / (K / 64):.0f} per k tile (2048 = MFMA bound), "
                         f"{ticks * 10} ns -> {cyc / max(ticks, 1) * 100:.0f} MHz")
            if int(os.environ.get("SPN_GEMM_DBG", "0")) & 64:
                extra = "  cycles (setup / asm or prologue / k loop / epilogue):"
                bias = torch.randn(N, device="cuda")
                for name, kw in (("bf16", {}), ("bf16+bias", dict(bias=bias)),
                                 ("gelu+pre", dict(bias=bias, act=ops.ACT_QUICKGELU, want_pre=True))):
                    out = ops.gemm_nt(a, b, **kw); torch.cuda.synchronize()
                    out = out[0] if isinstance(out, tuple) else out
                    c = [x & 0xffffffff for x in out.view(-1)[:8].view(torch.int32).cpu().tolist()[:4]]
                    extra += f" {name}: {c[0]}/{c[1]}/{c[3]}/{c[2]};"
            print(f"  {M}x{N}x{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF{extra}", flush=True)
else:
    if sys.argv[1:2] == ["ablate"]:
        for var in ["0", "1", "2", "3", "4"]:
            print(f"== SPN_GEMM_CFG=7 SPN_NT3_VAR={var} (1 no DMA, 2 no fragment reads, 3 no barrier, 4 MFMA only)", flush=True)
            env = dict(os.environ, SPN_GEMM_CFG="7", SPN_NT3_VAR=var, SPN_GEMM_DBG="32")
            subprocess.run(["timeout", "120", sys.executable, __file__, "child", "time"], env=env)
        sys.exit(0)
    if sys.argv[1:2] == ["phases2"]:          # the production 8-wave kernel: setup / prologue / k loop / epilogue
        env = dict(os.environ, SPN_GEMM_DBG="64")
        subprocess.run(["timeout", "120", sys.executable, __file__, "child", "time"], env=env)
        sys.exit(0)
    if sys.argv[1:2] == ["phases"]:
        env = dict(os.environ, SPN_GEMM_CFG="7", SPN_GEMM_DBG="64")
        subprocess.run(["timeout", "120", sys.executable, __file__, "child", "time"], env=env)
        sys.exit(0)
    for cfg in (sys.argv[1:] or ["7", "3"]):
        print(f"== SPN_GEMM_CFG={cfg}", flush=True)
        env = dict(os.environ, SPN_GEMM_CFG=cfg)
        subprocess.run(["timeout", "120", sys.executable, __file__, "child", "check"], env=env)
        subprocess.run(["timeout", "120", sys.executable, __file__, "child", "time"], env=env)
