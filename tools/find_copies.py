"""Which host lines launch kernels that are not this library's inside one product step?  Runs the config-2 trainer step under
torch.profiler (with Python stacks) and prints every device activity whose name does not start with spn:: - rocclr copy /
fill kernels, at::native kernels, memcpy / memset - with the innermost repo frame of its launching call.
    python tools/find_copies.py [--blip] [--packed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch.profiler import ProfilerActivity, profile
    from spn4cir_amd import synthetic
    from spn4cir_amd.models import CIRPlus
    from spn4cir_amd.trainer import Stage2Trainer
    dev = torch.device("cuda", 0)
    if "--blip" in sys.argv:
        return blip(dev)
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
    model = CIRPlus(synthetic.text_state_dict(W, layers, D, seed=0), tau=0.02, device=dev, plus=True)
    target, refer = synthetic.banks(40000, D, seed=2)
    tr = Stage2Trainer(model, lr=2e-5)
    tr.set_banks(refer, target)
    B = 256
    ids_host = synthetic.token_ids(B, seed=1)
    ids = ids_host.to(dev)
    ridx, lab = [x.to(dev) for x in synthetic.triplet_indices(B, 40000, seed=4)]
    kw = {}
    if "--packed" in sys.argv:
        cu, total = tr.tower.cu_seqlens(ids_host)
        kw = dict(cu_seqlens=cu.to(dev), total_rows=total)
    for _ in range(3):
        tr.step(ids, ridx, lab, **kw)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            tr.step(ids, ridx, lab, **kw)
        torch.cuda.synchronize()
    report(prof)


def blip(dev):
    """The blip4cir stage-2 step (config 4's shape at a small token bank) under the same lens."""
    from torch.profiler import ProfilerActivity, profile
    from spn4cir_amd.fusion import BlipStage2Trainer, FusionEncoder
    g = torch.Generator().manual_seed(0)
    enc = FusionEncoder(768, 12, 12, 3072, 768, 256, 30524, 512, dev)
    with torch.no_grad():
        for k, v in enc.named_views().items():
            if k.endswith("LayerNorm.weight"):
                v.fill_(1.0)
            elif v.dim() >= 2:
                v.copy_((torch.randn(v.shape, generator=g) * 0.02).to(dev))
    enc.mark_stale()
    B, L, S, M, N = 128, 32, 577, 30000, 500
    ids = torch.randint(1000, 30522, (B, L), generator=g, dtype=torch.int32)
    ids[:, 0] = 30523
    mask = torch.ones(B, L, dtype=torch.int32)
    ids, mask = ids.to(dev), mask.to(dev)
    bank = torch.randn(N, S, 768, device=dev).to(torch.bfloat16)
    ridx = torch.randint(0, N, (B,), generator=g).to(dev)
    labels = torch.randint(0, M, (B,), generator=g).to(dev)
    tr = BlipStage2Trainer(enc, tau=0.03, lr=5e-6, bank_mode="replicated")
    tr.set_bank(torch.nn.functional.normalize(torch.randn(M, 256, generator=g)))
    tr.set_token_bank(bank)
    for _ in range(3):
        tr.step(ids, mask, None, labels, token_idx=ridx)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            tr.step(ids, mask, None, labels, token_idx=ridx)
        torch.cuda.synchronize()
    report(prof)


def report(prof):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CUDA:
            continue
        name = ev.name
        if "spn" in name[:12] or name.startswith("_ZN3spn"):
            continue
        seen.setdefault(name[:90], []).append(ev)
    # map device activities back to their launching CPU op through the correlation: print CPU-side ops with stacks instead
    print("non-spn device activities in 2 steps:")
    for k, v in sorted(seen.items(), key=lambda t: -len(t[1])):
        print(f"  {len(v):4d} x {k}   total {sum(e.device_time_total for e in v):.1f} us")
    print("\nCPU-side aten ops / runtime calls in 2 steps (with the innermost repo frame):")
    agg = {}
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            continue
        if not (ev.name.startswith("aten::") or "Memcpy" in ev.name or "Memset" in ev.name):
            continue
        if ev.name in ("aten::empty", "aten::view", "aten::reshape", "aten::as_strided", "aten::empty_strided", "aten::slice",
                       "aten::select", "aten::unsqueeze", "aten::empty_like", "aten::_unsafe_view", "aten::alias", "aten::detach",
                       "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero", "aten::expand"):
            continue
        frame = next((s for s in (ev.stack or []) if root in s and "find_copies" not in s), (ev.stack or ["?"])[0] if ev.stack else "?")
        agg.setdefault((ev.name, frame), 0)
        agg[(ev.name, frame)] += 1
    for (n, f), c in sorted(agg.items(), key=lambda t: -t[1]):
        print(f"  {c:4d} x {n:28s} {f}")


if __name__ == "__main__":
    main()
