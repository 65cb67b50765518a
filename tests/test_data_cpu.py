"""Host data layer against structures captured from the reference's CIRDataset on the same synthetic trees
(tests/golden/make_golden_data.py): name -> id numbering, optimized_images.json override, unlabeled enumeration,
the item protocol of every split, and the FashionIQ caption policy on pinned and seeded draws."""
import json
import os
import random

import pytest


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "data_layer.json")) as f:
        return json.load(f)


def _snapshot(ds, root, n_items=6, seed=5):
    out = {"len": len(ds)}
    for k in ("targetname2id", "imagename2id", "imagenames", "imagepaths", "image_names", "unlabeled_imagenames"):
        if hasattr(ds, k):
            v = getattr(ds, k)
            out[k] = [str(x) for x in v] if isinstance(v, list) else v
    if ds.mode == "relative" and (ds.split != "train" or ds.use_bank):
        random.seed(seed)
        out["items"] = [list(ds[i]) for i in range(min(n_items, len(ds)))]
    return json.loads(json.dumps(out).replace(root, "<ROOT>"))


def test_caption_policy(golden):
    from spn4cir_amd.data import generate_randomized_fiq_caption
    caps = ["is red.", " has long sleeves ?"]
    assert [generate_randomized_fiq_caption(caps, type=t) for t in range(4)] == golden["caption_types"]
    draws = []
    for s in range(40):
        random.seed(s)
        draws.append(generate_randomized_fiq_caption(caps))
    assert draws == golden["caption_draws"]

    class Edge:                      # a draw exactly on a boundary falls through to the last form (strict inequalities)
        def __init__(self, u):
            self.u = u

        def random(self):
            return self.u
    assert generate_randomized_fiq_caption(caps, rng=Edge(0.25)) == "has long sleeves"
    assert generate_randomized_fiq_caption(caps, rng=Edge(0.5)) == "has long sleeves"


@pytest.mark.parametrize("override", [False, True])
def test_datasets_match_reference(golden, golden_dir, tmp_path, override):
    import sys
    sys.path.insert(0, golden_dir)
    import data_tree
    from spn4cir_amd.data import CIRDataset
    root = str(tmp_path / f"o{int(override)}")
    fiq, cirr = data_tree.build(root, seed=0, with_override=override)
    cases = [("fiq", "train", "relative", fiq, dict(plus=True)), ("fiq", "train", "relative", fiq, dict(dress_types=["shirt"])),
             ("fiq", "val", "relative", fiq, {}), ("fiq", "train", "unlabeled", fiq, {}),
             ("cirr", "train", "relative", cirr, dict(plus=True)), ("cirr", "val", "relative", cirr, {}),
             ("cirr", "test1", "relative", cirr, {})]
    for name, split, mode, path, kw in cases:
        ds = CIRDataset(name, split, mode, lambda im: im, data_path=path, **kw)
        if split == "train":
            ds.use_bank = True
        key = f"o{int(override)}/{name}/{split}/{mode}/{json.dumps(kw, sort_keys=True)}"
        got, ref = _snapshot(ds, root), golden[key]
        for k in ref:
            assert got.get(k) == ref[k], (key, k)
