"""Helpers the reference's scripts import from clip4cir/utils.py."""
from pathlib import Path

import torch

from .preprocess import gpu_decode_scope, realize_items, stack_images


def _chunks(dataset, bs=128, decode_bs=1024):
    """(name, image) items in batches of `bs`; fetched `decode_bs` at a time so that undecoded JPEGs (file bytes under
    gpu_decode_scope) are decoded on the GPU a whole chunk per call (a lane per file)."""
    n = len(dataset)
    for s in range(0, n, decode_bs):
        with gpu_decode_scope(dataset):
            items = [dataset[i] for i in range(s, min(n, s + decode_bs))]
        items = realize_items([it for it in items if it is not None])
        for k in range(0, len(items), bs):
            yield items[k:k + bs]


def extract_index_features(dataset, model, device=torch.device("cuda")):
    """utils.py:24-50: (names, images) items of a 'classic' dataset -> (features [N, D], names)."""
    feats, names = [], []
    for items in _chunks(dataset):
        names.extend(it[0] for it in items)
        feats.append(model.encode_image(stack_images([it[1] for it in items])))
    return torch.vstack(feats), names


def extract_index_features_fusion(dataset, model, device=torch.device("cuda")):
    """tgcir/utils.py:24-51 (= blip4cir/utils.py): -> (token features [N, T, C] on the CPU, pooled + normalised features
    [N, C] on the device, names) from model.img_embed(images, return_pool_and_normalized=True)."""
    toks, pooled, names = [], [], []
    for items in _chunks(dataset):
        names.extend(it[0] for it in items)
        t, p = model.img_embed(stack_images([it[1] for it in items]), return_pool_and_normalized=True)
        toks.append(t.cpu())
        pooled.append(p)
    return torch.vstack(toks), torch.vstack(pooled), names


def save_model(name, cur_epoch, model_to_save, training_path):
    """utils.py:53-67: {'epoch', 'state_dict'} at <training_path>/<name>.pt"""
    path = Path(training_path)
    path.mkdir(exist_ok=True, parents=True)
    torch.save({"epoch": cur_epoch, "state_dict": model_to_save.state_dict()}, str(path / f"{name}.pt"))


class RunningAverage:
    """utils.py:70-91."""

    def __init__(self):
        self.steps, self.total = 0, 0

    def update(self, val):
        self.total += val
        self.steps += 1

    def __call__(self):
        return self.total / float(self.steps)
