"""Host-side data layer of the stage-2 scripts: the index structures and item
protocol of `CIRDataset` (clip4cir/data_utils_negplus.py:121-340) and the FashionIQ caption policy
(:100-118).  Pure host logic - file lists, name -> id numbering, caption choice; images are only opened and
handed to `preprocess` (e.g. the GPU TargetPadTransform when used from the main process).

What the training loop relies on (train_negplus.py:93-111, models_negplus.py:59-125):
  * `targetname2id`: targets numbered in order of first appearance over the triplets;
  * `imagename2id` / `imagenames` / `imagepaths`: every reference and target image, numbered in order of first
    appearance (reference before target inside a triplet), optionally replaced wholesale by
    `<data_path>/optimized_images.json` = [imagenames, imagepaths, imagename2id];
  * items in `use_bank` mode: (caption, triplet index, target_index, target_index_all, reference_index_all).
"""
import json
import os
import random

import PIL.Image
from torch.utils.data import Dataset

_STRIP = ".?, "
_FIQ_TYPES = ("dress", "shirt", "toptee")


def generate_randomized_fiq_caption(captions, type=-1, rng=random):
    """Four-way FashionIQ caption policy: "a and b" / "b and a" / "a" / "b" on a uniform draw u.
    The reference compares with strict inequalities on both sides (0.25 < u < 0.5, 0.5 < u < 0.75), so a draw
    that lands exactly on 0.25 or 0.5 falls through to the last form; `type` 0..3 pins the form."""
    u = {0: 0.12, 1: 0.37, 2: 0.62, 3: 0.88}.get(type)
    if u is None:
        u = rng.random()
    a, b = captions[0].strip(_STRIP), captions[1].strip(_STRIP)
    if u < 0.25:
        return f"{a} and {b}"
    if 0.25 < u < 0.5:
        return f"{b} and {a}"
    if 0.5 < u < 0.75:
        return a
    return b


def _load(path):
    with open(path) as f:
        return json.load(f)


def _fiq_triplets(root, split, dress_types, plus):
    cap, img = os.path.join(root, "captions"), os.path.join(root, "images")
    raw = []
    for t in dress_types:
        raw += _load(os.path.join(cap, f"cap.{t}.{split}.json"))
    if split == "train" and plus:
        raw += _load(os.path.join(cap, "cap.extend_clip.train.json"))
    return [dict(reference=os.path.join(img, r["candidate"] + ".png"), reference_name=r["candidate"],
                 target=os.path.join(img, r["target"] + ".png"), target_name=r["target"], captions=r["captions"])
            for r in raw], img


def _cirr_triplets(root, split, plus):
    cap = os.path.join(root, "cirr/captions")
    relpath = _load(os.path.join(root, "cirr/image_splits", f"split.rc2.{split}.json"))
    raw = _load(os.path.join(cap, f"cap.rc2.{split}.json"))
    if split == "train" and plus:
        raw += _load(os.path.join(cap, "cap.rc2.train.extend_clip.json"))
    out = []
    for r in raw:
        hard = r.get("target_hard")
        out.append(dict(reference=os.path.join(root, relpath[r["reference"]]), reference_name=r["reference"],
                        target=os.path.join(root, relpath[hard]) if hard is not None else "",
                        target_name=hard if hard is not None else "",
                        captions=[r["caption"]] if isinstance(r["caption"], str) else r["caption"],
                        pairid=r["pairid"], group_members=r["img_set"]["members"]))
    return out, relpath


def number_images(triplets):
    """-> (targetname2id, imagename2id, imagenames) in first-appearance order."""
    tgt, img, names = {}, {}, []
    for t in triplets:
        r, g = t["reference_name"], t["target_name"]
        tgt.setdefault(g, len(tgt))
        for n in (r, g):
            if n not in img:
                img[n] = len(names)
                names.append(n)
    return tgt, img, names


class CIRDataset(Dataset):
    def __init__(self, data_name, split, mode, preprocess, data_path="./", dress_types=None, val_ret_train=False,
                 fiq_val_type=0, plus=False, coco_root=None):
        dress_types = list(_FIQ_TYPES) if dress_types is None else list(dress_types)
        if any(t not in _FIQ_TYPES for t in dress_types):
            raise AssertionError(f"dress_types must be among {_FIQ_TYPES}")
        if data_name not in ("fiq", "cirr"):
            raise ValueError(data_name)
        self.data_name, self.split, self.mode, self.preprocess = data_name, split, mode, preprocess
        self.data_path, self.dress_types = data_path, dress_types
        self.val_ret_train, self.fiq_val_type = val_ret_train, fiq_val_type
        self.use_bank = False
        self.targetname2id, self.imagename2id, self.imagenames, self.imagepaths = {}, {}, [], []
        if data_name == "fiq":
            self.triplets, self.image_path = _fiq_triplets(data_path, split, dress_types, plus)
            self.image_names = []
            for t in dress_types:
                self.image_names += _load(os.path.join(data_path, "image_splits", f"split.{t}.{split}.json"))
            self.val_image_names = []
            if fiq_val_type == 1 and split == "val":
                seen = set()
                for t in self.triplets:
                    seen.update((t["reference_name"], t["target_name"]))
                self.val_image_names = list(seen)
        else:
            self.triplets, self.name_to_relpath = _cirr_triplets(data_path, split, plus)
            self.image_path = data_path
            self._cirr_root = coco_root if coco_root is not None else data_path
        if split == "train":
            self.targetname2id, self.imagename2id, self.imagenames = number_images(self.triplets)
            self.imagepaths = [self._path_of(n) for n in self.imagenames]
            override = os.path.join(data_path, "optimized_images.json")
            if os.path.exists(override):
                self.imagenames, self.imagepaths, self.imagename2id = _load(override)
            self.target_id, self.image_id = len(self.targetname2id), len(self.imagenames)
        if mode == "unlabeled":
            self.unlabeled_imagenames = self._unlabeled()

    def _path_of(self, name):
        if self.data_name == "fiq":
            return os.path.join(self.image_path, name + ".png")
        return os.path.join(self.image_path, self.name_to_relpath[name])

    def _unlabeled(self):
        if self.data_name == "fiq":
            return [os.path.join(self.image_path, n + ".png") for n in self.image_names if n not in self.imagename2id]
        labelled = set(self.imagenames)
        out = [os.path.join(self._cirr_root, "cirr_dataset", rel) for n, rel in self.name_to_relpath.items()
               if n not in labelled]
        return out + list(_load(os.path.join(self.data_path, "coco_image.json")))

    def _open(self, path):
        return self.preprocess(PIL.Image.open(path))

    def _train_caption(self, captions):
        if len(captions) == 1:
            return captions[0]
        return generate_randomized_fiq_caption(captions) if self.data_name == "fiq" else random.choice(captions)

    def __len__(self):
        if self.mode == "relative":
            return len(self.triplets)
        if self.mode == "unlabeled":
            return len(self.unlabeled_imagenames)
        if self.data_name == "cirr":
            return len(self.name_to_relpath)
        return len(self.image_names if self.fiq_val_type == 0 else self.val_image_names)

    def __getitem__(self, index):
        if self.mode == "unlabeled":
            return self._open(self.unlabeled_imagenames[index])
        if self.mode == "classic":
            if self.data_name == "cirr":
                name = list(self.name_to_relpath.keys())[index]
                return name, self._open(os.path.join(self._cirr_root, "cirr_dataset", self.name_to_relpath[name]))
            if self.fiq_val_type == 1:
                assert self.split == "val"
                name = self.val_image_names[index]
            else:
                name = self.image_names[index]
            return name, self._open(os.path.join(self.image_path, name + ".png"))
        t = self.triplets[index]
        caps = t["captions"]
        if self.split == "train":
            caption = self._train_caption(caps)
            ids = (self.targetname2id[t["target_name"]], self.imagename2id[t["reference_name"]],
                   self.imagename2id[t["target_name"]])
            if self.use_bank:
                return caption, index, ids[0], ids[2], ids[1]
            return self._open(t["reference"]), caption, self._open(t["target"]), index, ids[0], ids[1], ids[2]
        if self.split == "val" and self.val_ret_train:
            caption = generate_randomized_fiq_caption(caps, type=0) if len(caps) > 1 else caps[0]
            return self._open(t["reference"]), caption, self._open(t["target"])
        if self.split == "val":
            if self.data_name == "fiq":
                return t["reference_name"], t["target_name"], caps
            return t["reference_name"], t["target_name"], caps[0], t["group_members"]
        if self.split == "test1":
            assert self.data_name == "cirr"
            return t["pairid"], t["reference_name"], caps[0], t["group_members"]
        raise ValueError(f"unsupported split {self.split!r} in relative mode")
