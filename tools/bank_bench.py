"""Bank InfoNCE kernels alone (dev tool): forward / backward time and the achieved fraction of the HBM roofline for the
shapes of BASELINE configs 2, 3 and 5: the recomputing two-pass kernels, the saved pair, and the fused single pass (one
child process each).

    python tools/bank_bench.py [--rotate 8]

--rotate n: n copies of the bank, used round robin, so that a pass cannot be served from the 256 MB Infinity Cache (a
40 000 x 768 bf16 bank is 61 MB and would otherwise be cache-resident between back-to-back launches; in the training step
13 ms of other traffic evict it)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(32, 40000, 768, "bf16"), (256, 40000, 768, "bf16"), (32, 100000, 768, "bf16"), (32, 100000, 768, "fp8"),
         (16, 400000, 768, "bf16"), (16, 400000, 768, "fp8"), (64, 40000, 768, "bf16"), (256, 100000, 768, "fp8"),
         (32, 30000, 256, "bf16"), (128, 30000, 256, "bf16")]
if os.environ.get("BANK_BENCH_QUICK"):
    CASES = [(32, 40000, 768, "bf16"), (16, 400000, 768, "bf16"), (16, 400000, 768, "fp8")]


def child(rotate):
    sys.path.insert(0, ROOT)
    import torch
    from spn4cir_amd import ops
    out = []
    for B, M, D, dt in CASES:
        g = torch.Generator().manual_seed(0)
        banks = []
        for r in range(rotate):
            bank = torch.nn.functional.normalize(torch.randn(M, D, generator=g)).cuda()
            banks.append(ops.prepare_bank(bank, dt))
        q = torch.nn.functional.normalize(torch.randn(B, D, generator=g)).cuda()
        _, qb, _ = ops.combine_l2norm_fwd(None, None, q)
        labels = torch.randint(0, M, (B,), generator=g).cuda()
        save = ops.bank_logits_buffer(B, M, "cuda") if os.environ.get("SPN_BANK_SAVE") == "1" else None
        kw = {"save": save} if save is not None else {}
        kwb = {"saved": save} if save is not None else {}
        stats = ops.bank_stats_fwd(qb, banks[0], labels, 50.0, **kw)
        lse, _, _ = ops.bank_loss_finalize(stats, M)

        def timed(fn, n=24, warm=4):
            for i in range(warm):
                fn(i)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(n):
                fn(i)
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / n * 1e-3

        tf = timed(lambda i: ops.bank_stats_fwd(qb, banks[i % rotate], labels, 50.0, **kw))
        tb = timed(lambda i: ops.bank_grad_q(qb, banks[i % rotate], labels, 50.0, lse, 1.0 / B, **kwb))
        # the main kernel alone (HIP events recorded by the library around its launch; the fold launches are outside)
        import ctypes as C
        from spn4cir_amd import _lib
        L = _lib.lib()
        L.spn_prof_enable(256)
        L.spn_prof_select(0x30, 1)
        for i in range(16):
            ops.bank_stats_fwd(qb, banks[i % rotate], labels, 50.0, **kw)
            ops.bank_grad_q(qb, banks[i % rotate], labels, 50.0, lse, 1.0 / B, **kwb)
        torch.cuda.synchronize()
        kt = []
        for kid in (4, 5):
            ms, work, n = C.c_double(), C.c_double(), C.c_int()
            L.spn_prof_collect(kid, C.byref(ms), C.byref(work), C.byref(n))
            kt.append(ms.value / max(1, n.value) * 1e-3)
        L.spn_prof_disable()
        L.spn_prof_reset()
        tf, tb, tf_all, tb_all = kt[0], kt[1], tf, tb
        # the whole loss step in one call (spn_bank_step: the pass + ONE tail launch), host-timed, where it is served
        ts = None
        if save is not None and ops.bank_step_ok(B, M, qb.shape[1], banks[0]):
            ts = timed(lambda i: ops.bank_step(qb, banks[i % rotate], labels, 50.0, 1.0 / B, save))
        eb = 1 if dt == "fp8" else 2
        bytes_pass = M * D * eb
        out.append(dict(B=B, M=M, D=D, bank=dt, fwd_us=round(tf * 1e6, 1), bwd_us=round(tb * 1e6, 1),
                        fwd_call_us=round(tf_all * 1e6, 1), bwd_call_us=round(tb_all * 1e6, 1),
                        fwd_TBps=round(bytes_pass / tf / 1e12, 2), bwd_TBps=round(bytes_pass / tb / 1e12, 2),
                        pair_frac_of_8TBps=round(2 * bytes_pass / (tf + tb) / 8e12, 3),
                        step_call_us=None if ts is None else round(ts * 1e6, 1)))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rotate", type=int, default=8)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.rotate)
    res = {}
    configs = (("recompute", {"SPN_BANK2": "0", "SPN_BANK_SAVE": "0"}),
               ("saved", {"SPN_BANK2": os.environ.get("BANK_BENCH_GEN2", "0"), "SPN_BANK_FUSED": "0", "SPN_BANK_SAVE": "1"}),
               ("fused", {"SPN_BANK2": "0", "SPN_BANK_FUSED_LARGE": "1", "SPN_BANK_SAVE": "1"}))
    for name, env in configs:
        p = subprocess.run([sys.executable, __file__, "--child", "--rotate", str(a.rotate)], env=dict(os.environ, **env),
                           capture_output=True, text=True)
        if p.returncode:
            print(p.stdout[-2000:], p.stderr[-4000:])
            raise SystemExit(1)
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    print(f"bank InfoNCE passes: the main kernels alone (HIP events recorded by the library around each launch, 16 launches "
          f"rotating over {a.rotate} bank copies); call = whole op from the host incl. fold launches and wrapper overhead.\n"
          f"recompute = two passes, logits recomputed in the backward pass; saved = two passes, the backward pass reads what the "
          f"forward pass kept (B >= 256: p + G^T + TN GEMM; below: BANK_BENCH_GEN2=1 for the streaming pair, else = recompute);\n"
          f"fused = ONE pass over the bank (statistics + unnormalised dq), the backward call folds the chunk partials.")
    print(f"{'B':>4} {'M':>7} {'D':>5} {'bank':>5} | {'recompute f':>11} {'b':>6} {'sum':>6} | {'saved f':>8} {'b':>6} {'sum':>6} | "
          f"{'fused pass':>10} {'fold':>6} {'sum':>6} | {'pass: one read, TB/s':>20} {'/8':>5}")
    for o, n, f in zip(res["recompute"], res["saved"], res["fused"]):
        eb = 1 if o["bank"] == "fp8" else 2
        tbps = o["M"] * o["D"] * eb / (f["fwd_us"] * 1e-6) / 1e12            # the bank is read once per step
        print(f"{o['B']:>4} {o['M']:>7} {o['D']:>5} {o['bank']:>5} | {o['fwd_us']:>11} {o['bwd_us']:>6} {o['fwd_us'] + o['bwd_us']:>6.1f} | "
              f"{n['fwd_us']:>8} {n['bwd_us']:>6} {n['fwd_us'] + n['bwd_us']:>6.1f} | {f['fwd_us']:>10} {f['bwd_us']:>6} "
              f"{f['fwd_us'] + f['bwd_us']:>6.1f} | {tbps:>20.2f} {tbps / 8:>5.2f}   call {o['fwd_call_us']}/{o['bwd_call_us']} -> "
              f"{f['fwd_call_us']}/{f['bwd_call_us']}   spn_bank_step (one call, host-timed back to back): {f.get('step_call_us')}")


if __name__ == "__main__":
    main()
