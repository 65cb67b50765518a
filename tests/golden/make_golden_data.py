"""Golden index structures / items of the reference's CIRDataset (clip4cir/data_utils_negplus.py) on the synthetic
trees of data_tree.py, plus its caption policy on pinned draws.  Build container only.

    python tests/golden/make_golden_data.py  ->  tests/golden/data_layer.json
"""
import json
import os
import random
import sys
import tempfile

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from make_golden import REF, install_stubs  # noqa: E402
import data_tree  # noqa: E402


def snapshot(ds, n_items=6, seed=5):
    out = {"len": len(ds)}
    for k in ("targetname2id", "imagename2id", "imagenames", "imagepaths", "image_names", "unlabeled_imagenames"):
        if hasattr(ds, k):
            v = getattr(ds, k)
            out[k] = [str(x) for x in v] if isinstance(v, list) else v
    if ds.mode == "relative" and (ds.split != "train" or ds.use_bank):
        random.seed(seed)
        out["items"] = [list(ds[i]) for i in range(min(n_items, len(ds)))]
    return out


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "clip4cir"))
    import data_utils_negplus as du  # noqa: E402
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for override in (False, True):
            root = os.path.join(tmp, f"o{int(override)}")
            fiq, cirr = data_tree.build(root, seed=0, with_override=override)
            cases = [("fiq", "train", "relative", fiq, dict(plus=True)), ("fiq", "train", "relative", fiq, dict(dress_types=["shirt"])),
                     ("fiq", "val", "relative", fiq, {}), ("fiq", "train", "unlabeled", fiq, {}),
                     ("cirr", "train", "relative", cirr, dict(plus=True)), ("cirr", "val", "relative", cirr, {}),
                     ("cirr", "test1", "relative", cirr, {})]
            for name, split, mode, path, kw in cases:
                ds = du.CIRDataset(name, split, mode, lambda im: im, data_path=path, **kw)
                if split == "train":
                    ds.use_bank = True
                key = f"o{int(override)}/{name}/{split}/{mode}/{json.dumps(kw, sort_keys=True)}"
                snap = snapshot(ds)
                res[key] = json.loads(json.dumps(snap).replace(root, "<ROOT>"))
    caps = ["is red.", " has long sleeves ?"]
    res["caption_types"] = [du.generate_randomized_fiq_caption(caps, type=t) for t in range(4)]
    draws = []
    for s in range(40):
        random.seed(s)
        draws.append(du.generate_randomized_fiq_caption(caps))
    res["caption_draws"] = draws
    with open(os.path.join(OUT, "data_layer.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print(len(res), "entries")


if __name__ == "__main__":
    main()
