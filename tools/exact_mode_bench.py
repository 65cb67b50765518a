import sys, time, torch
sys.path.insert(0, '.')
from spn4cir_amd import synthetic
from spn4cir_amd.text_tower import TextTower
from spn4cir_amd.vision_tower import VisionTower
W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-L/14"]
t = TextTower(W, layers, heads, D, 49408, 77, "cuda"); t.load_clip_state_dict(synthetic.text_state_dict(W, layers, D, seed=0))
ids = synthetic.token_ids(256, seed=1).cuda()
for f, name in ((t.forward, "bf16"), (t.forward_exact, "exact")):
    f(ids); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f(ids)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"text ViT-L/14 B=256 {name}: {dt*1e3:.1f} ms  {256/dt:.0f} captions/s")
v = VisionTower(1024, 24, 16, 14, 224, 768, "cuda")
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for k, x in v.named_views().items():
        if x.dim() >= 2: x.copy_((torch.randn(x.shape, generator=g) * 0.02).cuda())
        else: x.fill_(1.0 if ("weight" in k or k.startswith("ln_")) else 0.0)
v.mark_stale()
img = torch.randn(64, 3, 224, 224, generator=g).cuda()
for f, name in ((v.forward, "bf16"), (v.forward_exact, "exact")):
    f(img); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): f(img)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"vision ViT-L/14 B=64 {name}: {dt*1e3:.1f} ms  {64/dt:.0f} images/s")
