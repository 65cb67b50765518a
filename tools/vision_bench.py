"""Image-tower forward throughput (bank extraction and validation are bound by it once
decode is off the critical path).  Random weights / images.

    python tools/vision_bench.py [--model ViT-L/14|ViT-B/32|ViT-B/16|BLIP-B/16-384] [--batch 256]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spn4cir_amd.vision_tower import VisionTower

CFG = {  # width, layers, heads, patch, res, embed_dim, kind
    "ViT-B/32": (768, 12, 12, 32, 224, 512, 0), "ViT-B/16": (768, 12, 12, 16, 224, 512, 0),
    "ViT-L/14": (1024, 24, 16, 14, 224, 768, 0), "BLIP-B/16-384": (768, 12, 12, 16, 384, 256, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    W, layers, H, p, res, D, kind = CFG[a.model]
    t = VisionTower(W, layers, H, p, res, D, "cuda", kind=kind)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for k, v in t.named_views().items():
            if v.dim() >= 2:
                v.copy_((torch.randn(v.shape, generator=g) * 0.02).cuda())
            elif "weight" in k or k.startswith("ln_") or "norm" in k:
                v.fill_(1.0)
    t.mark_stale()
    img = torch.randn(a.batch, 3, res, res, generator=g).cuda()
    for _ in range(2):
        t.forward(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t.forward(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    S = (res // p) ** 2 + 1
    per_layer = 2 * S * W * 3 * W + 4 * S * S * W + 2 * S * W * W + 4 * S * W * 4 * W
    flops = layers * per_layer + 2 * (S - 1) * 3 * p * p * W + 2 * W * D
    print(json.dumps({"model": a.model, "batch": a.batch, "tokens": S, "images_per_s": round(a.batch / dt, 1),
                      "ms_per_batch": round(dt * 1e3, 2), "gflop_per_image": round(flops / 1e9, 1),
                      "tflops": round(flops * a.batch / dt / 1e12, 1)}))


if __name__ == "__main__":
    main()
