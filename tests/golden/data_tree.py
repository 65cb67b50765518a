"""Seeded synthetic FashionIQ / CIRR directory trees (JSON only + 1x1 PNGs), shared by the golden generator and
the tests: the reference's CIRDataset and spn4cir_amd.data.CIRDataset are both pointed at the same tree."""
import json
import os
import random

from PIL import Image


def _dump(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(obj, f)


def build(root, seed=0, with_override=False):
    rng = random.Random(seed)
    fiq = os.path.join(root, "fiq")
    words = ["red", "longer", "striped", "sleeveless", "darker", "floral", "shorter", "plain"]
    for t in ("dress", "shirt", "toptee"):
        names = [f"{t[0].upper()}{i:04d}" for i in range(14)]
        for split in ("train", "val"):
            pool = names[:9] if split == "train" else names[7:]
            trip = []
            for _ in range(8):
                c, g = rng.sample(pool, 2)
                trip.append({"candidate": c, "target": g,
                             "captions": [f"is {rng.choice(words)}.", f"has {rng.choice(words)} pattern ?"]})
            _dump(os.path.join(fiq, "captions", f"cap.{t}.{split}.json"), trip)
            _dump(os.path.join(fiq, "image_splits", f"split.{t}.{split}.json"), pool + [f"{t}_extra_{split}_{k}" for k in range(3)])
        for n in names:
            os.makedirs(os.path.join(fiq, "images"), exist_ok=True)
            Image.new("RGB", (1, 1), (1, 2, 3)).save(os.path.join(fiq, "images", n + ".png"))
    _dump(os.path.join(fiq, "captions", "cap.extend_clip.train.json"),
          [{"candidate": "D0001", "target": "S0002", "captions": ["is plain", "is red"]}])
    cirr = os.path.join(root, "cirr_root")
    names = [f"img-{i}" for i in range(16)]
    rel = {n: f"./train/{i % 3}/{n}.png" for i, n in enumerate(names)}
    for split in ("train", "val", "test1"):
        _dump(os.path.join(cirr, "cirr/image_splits", f"split.rc2.{split}.json"), rel)
        trip = []
        for k in range(9):
            r, g = rng.sample(names[:10], 2)
            row = {"reference": r, "caption": f"make it {rng.choice(words)}", "pairid": 100 + k,
                   "img_set": {"members": rng.sample(names, 5)}}
            if split != "test1":
                row["target_hard"] = g
            trip.append(row)
        _dump(os.path.join(cirr, "cirr/captions", f"cap.rc2.{split}.json"), trip)
    _dump(os.path.join(cirr, "cirr/captions", "cap.rc2.train.extend_clip.json"),
          [{"reference": "img-12", "target_hard": "img-13", "caption": ["a", "b"], "pairid": 999,
            "img_set": {"members": names[:5]}}])
    _dump(os.path.join(cirr, "coco_image.json"), ["/coco/a.jpg", "/coco/b.jpg"])
    if with_override:
        # a de-duplicated image list as deduplicate_images.py writes it: every train image name is mapped, but
        # the last few share the id (and file) of an earlier image
        seen = []
        for t in ("dress", "shirt", "toptee"):
            with open(os.path.join(fiq, "captions", f"cap.{t}.train.json")) as f:
                for r in json.load(f):
                    for n in (r["candidate"], r["target"]):
                        if n not in seen:
                            seen.append(n)
        for n in ("D0001", "S0002"):
            if n not in seen:
                seen.append(n)
        keep = sorted(seen)[:-3]
        mapping = {n: (keep.index(n) if n in keep else i % len(keep)) for i, n in enumerate(sorted(seen))}
        _dump(os.path.join(fiq, "optimized_images.json"),
              [keep, [os.path.join("/dedup", n + ".png") for n in keep], mapping])
    return fiq, cirr
