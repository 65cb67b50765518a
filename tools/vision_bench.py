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


def resnet_state_dict(layers, width, res, embed_dim, seed=0):
    """Seeded random ModifiedResNet weights with the shapes of clip/model.py:94-137."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, co, ci, k):
        sd[name] = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.05 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 0.8 + 0.4 * torch.rand(c, generator=g)

    conv("visual.conv1.weight", width // 2, 3, 3); bn("visual.bn1", width // 2)
    conv("visual.conv2.weight", width // 2, width // 2, 3); bn("visual.bn2", width // 2)
    conv("visual.conv3.weight", width, width // 2, 3); bn("visual.bn3", width)
    inpl = width
    for li, n in enumerate(layers, start=1):
        planes = width * 2 ** (li - 1)
        for bi in range(n):
            p = f"visual.layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            conv(p + "conv1.weight", planes, inpl, 1); bn(p + "bn1", planes)
            conv(p + "conv2.weight", planes, planes, 3); bn(p + "bn2", planes)
            conv(p + "conv3.weight", planes * 4, planes, 1); bn(p + "bn3", planes * 4)
            if stride > 1 or inpl != planes * 4:
                conv(p + "downsample.0.weight", planes * 4, inpl, 1); bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    C, grid = width * 32, res // 32
    sd["visual.attnpool.positional_embedding"] = torch.randn(grid * grid + 1, C, generator=g) / C ** 0.5
    for n, o in (("q", C), ("k", C), ("v", C), ("c", embed_dim)):
        sd[f"visual.attnpool.{n}_proj.weight"] = torch.randn(o, C, generator=g) * C ** -0.5
        sd[f"visual.attnpool.{n}_proj.bias"] = torch.zeros(o)
    return sd


RN_CFG = {"RN50": ((3, 4, 6, 3), 64, 224, 1024), "RN50x4": ((4, 6, 10, 6), 80, 288, 640)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--fast", action="store_true", help="ModifiedResNet towers: the bf16 throughput mode")
    a = ap.parse_args()
    if a.model in RN_CFG:
        from spn4cir_amd.resnet_tower import ResNetTower
        layers, width, res, D = RN_CFG[a.model]
        t = ResNetTower(resnet_state_dict(layers, width, res, D), "cuda", fast=a.fast)
        img = torch.randn(a.batch, 3, res, res).cuda()
        t.forward(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            t.forward(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(json.dumps({"model": a.model + (" (ModifiedResNet, bf16 fast path)" if a.fast else " (ModifiedResNet, fp32 path)"),
                          "batch": a.batch,
                          "images_per_s": round(a.batch / dt, 1), "ms_per_batch": round(dt * 1e3, 2)}))
        return
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
