"""Batched GPU JPEG decode: host parsing / upload / kernel time per batch, for FashionIQ-sized files (400 x 600, ~45 KB).
    python tools/jpeg_bench.py [batch ...]          (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from PIL import Image
from spn4cir_amd import jpeg

rng = np.random.default_rng(0)
def photo(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3))
    for k in range(12):                                   # smooth blobs + edges + mild texture: a product photo's statistics
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(30, 200)
        col = rng.uniform(0, 255, 3)
        img += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))[..., None] * col
    img = img / img.max() * 255
    img[h // 3: h // 3 + 150, w // 4: w // 4 + 120] = rng.uniform(0, 255, 3)
    img += rng.normal(0, 3, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)
files = []
for k in range(16):
    buf = io.BytesIO()
    Image.fromarray(photo(600, 400)).save(buf, "JPEG", quality=90)
    files.append(buf.getvalue())
print("file size KB:", [len(f) // 1024 for f in files[:8]])
t0 = time.perf_counter()
for f in files: np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
print(f"Pillow, one host core: {16 / (time.perf_counter() - t0):.0f} images/s")
for n in [int(a) for a in sys.argv[1:]] or [32, 128, 256, 1024]:
    fl = (files * ((n + 15) // 16))[:n]
    jpeg.decode_batch(fl, "cuda"); torch.cuda.synchronize()
    t0 = time.perf_counter(); b = jpeg.Batch(fl); t_host = time.perf_counter() - t0
    t0 = time.perf_counter(); out, _ = jpeg.decode_batch(fl, "cuda"); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"batch {n:5d}: host parse + assemble {t_host * 1e3:7.1f} ms, whole call {t_all * 1e3:7.1f} ms -> {n / t_all:8.0f} images/s; segments {b.segs.shape[0]}")
