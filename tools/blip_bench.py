"""BLIP4CIR stage-2 step throughput (BASELINE.json config 4 shape on ONE GPU): BERT-base fusion encoder
(12 layers, 768, 12 heads, cross-attention over 577 image tokens), B=128, 32-token captions, 30 000 x 256 bank,
tau 0.03, AdamW; reference token bank resident on the device in bf16 ([--images, 577, enc_width]: 30 000 images = 26.6 / 35.4 GB),
gathered per step by the library (spn_fusion_fwd_bank).

    python tools/blip_bench.py [--enc-width 768|1024] [--steps 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 tools/blip_bench.py
        (config 4: data parallel, B triplets per GPU, bank sharded over the ranks)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd import ops
from spn4cir_amd.fusion import BlipStage2Trainer, FusionEncoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--len", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=577)
    ap.add_argument("--enc-width", type=int, default=768)
    ap.add_argument("--bank", type=int, default=30000)
    ap.add_argument("--images", type=int, default=30000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bank-mode", default="sharded", choices=["sharded", "replicated"])
    ap.add_argument("--dense", action="store_true", help="attention mask on the device: the padded B x L rows (no packing)")
    ap.add_argument("--min-len", type=int, default=6, help="caption lengths are uniform in [min-len, len]")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    g = torch.Generator().manual_seed(0)
    enc = FusionEncoder(768, 12, 12, 3072, a.enc_width, 256, 30524, 512, dev)
    with torch.no_grad():
        for k, v in enc.named_views().items():
            if k.endswith("LayerNorm.weight"):
                v.fill_(1.0)
            elif v.dim() >= 2:
                v.copy_((torch.randn(v.shape, generator=g) * 0.02).to(dev))
    enc.mark_stale()
    B, L = a.batch, a.len
    ids = torch.randint(1000, 30522, (B, L), generator=g, dtype=torch.int32)
    ids[:, 0] = 30523
    lens = torch.randint(a.min_len, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int32)
    ids = (ids * mask).to(dev)
    if a.dense:
        mask = mask.to(dev)          # a host mask (what the tokenizer returns) lets the encoder drop the padded rows
    ref_bank = torch.empty(a.images, a.tokens, a.enc_width, dtype=torch.bfloat16, device=dev)   # per-image token bank (models.py:76)
    dgen = torch.Generator(device=dev).manual_seed(5)
    for s0 in range(0, a.images, 1000):
        n = min(1000, a.images - s0)
        ref_bank[s0:s0 + n].copy_(torch.randn(n, a.tokens, a.enc_width, generator=dgen, device=dev))
    ridx = torch.randint(0, a.images, (B,), generator=g).to(dev)
    labels = torch.randint(0, a.bank, (B,), generator=g).to(dev)
    trainer = BlipStage2Trainer(enc, tau=0.03, lr=5e-6, bank_mode=a.bank_mode)
    trainer.set_bank(torch.nn.functional.normalize(torch.randn(a.bank, 256, generator=g)))
    trainer.set_token_bank(ref_bank)

    def one(i):
        return trainer.step(ids, mask, None, labels, token_idx=ridx)             # rows gathered on the device (models.py:98)

    for i in range(a.warmup):
        loss = one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = one(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    if rank != 0:
        dist.destroy_process_group()
        return
    T, TS, W, I, E = B * L, B * a.tokens, 768, 3072, a.enc_width
    fwd = 12 * (2 * T * W * 3 * W + 2 * T * W * W * 2 + 2 * TS * E * 2 * W + 2 * T * W * I * 2
                + 4 * B * 12 * L * L * 64 + 4 * B * 12 * L * a.tokens * 64)
    print(json.dumps({"rows": "dense" if a.dense else f"packed ({int(lens.sum())} of {B * L})",
                      "workload": f"blip4cir stage-2 step, BERT-base fusion, B={B}, L={L}, {a.tokens} image tokens, "
                                  f"enc_width {E}, bank {a.bank}x256, {world} GPU(s)", "triplets_per_s": round(B * world / dt, 1),
                      "ms_per_step": round(dt * 1e3, 2), "loss": round(loss.item(), 4),
                      "model_tflops_per_gpu": round((3 * fwd - 12 * 2 * TS * E * 2 * W) / dt / 1e12, 1), "params_M": round(enc.n_params / 1e6, 2)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
