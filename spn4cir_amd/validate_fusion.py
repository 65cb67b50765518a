"""Recall@K validation for the fusion-style second stages (blip4cir, tgcir) on the MI355X kernels.

Mirrors blip4cir/validate.py and tgcir/validate.py (the two files differ only in the fusion call and the feature
width): `compute_fiq_val_metrics` (:21-58), `generate_fiq_val_predictions` (:61-110), `compute_cirr_val_metrics`
(:133-195), `generate_cirr_val_predictions` (:197-244).  What differs from clip4cir/validate.py: the query is
`model.img_txt_fusion(reference TOKEN features, captions)`, the gallery is the pre-pooled, pre-normalised
`index_features_normed_pooled`, FashionIQ captions are joined with `.capitalize()`, and the FashionIQ ranking does
not drop the reference image.  Scores are accumulated in fp64 and only the top 50 are selected, as in validate.py.
"""
import inspect
from typing import List, Tuple

import torch

from . import ops
from .validate import _batches, _name_index


def _fuse(model, ref_feats, captions):
    """tgcir: img_txt_fusion(ref_token, mod); blip4cir: img_txt_fusion(r_image_embeds, t_image_embeds, text[, train])."""
    fn = model.img_txt_fusion
    try:
        n = len([p for p in inspect.signature(fn).parameters.values() if p.default is inspect.Parameter.empty])
    except (TypeError, ValueError):
        n = 2
    out = fn(ref_feats, None, captions) if n >= 3 else fn(ref_feats, captions)
    return out.to(torch.float32)


def _normalized(x, device):
    return ops.combine_l2norm_fwd(None, None, x.to(device, torch.float32).contiguous())[0]


def generate_fiq_val_predictions(model, relative_val_dataset, index_names: List[str], index_features: torch.Tensor,
                                 device=torch.device("cuda")):
    """-> (predicted_features [N, D], target_names)   blip4cir/validate.py:61-110"""
    name2idx = _name_index(index_names)
    preds, target_names = [], []
    for batch in _batches(relative_val_dataset):
        caps = [f"{b[2][0].strip('.?, ').capitalize()} and {b[2][1].strip('.?, ')}" for b in batch]
        ridx = torch.tensor([name2idx[str(b[0])] for b in batch], dtype=torch.int64, device=index_features.device)
        preds.append(_fuse(model, index_features[ridx].to(device), caps).to(device))
        target_names.extend(b[1] for b in batch)
    return torch.vstack(preds), target_names


def compute_fiq_val_metrics(relative_val_dataset, model, index_features: torch.Tensor,
                            index_features_normed_pooled: torch.Tensor, index_names: List[str],
                            device=torch.device("cuda")) -> Tuple[float, float]:
    """blip4cir/validate.py:21-58 -> (recall@10, recall@50) in percent; the reference image stays in the ranking."""
    predicted, target_names = generate_fiq_val_predictions(model, relative_val_dataset, index_names, index_features, device)
    name2idx = _name_index(index_names)
    gallery = index_features_normed_pooled.to(device, torch.float32).contiguous()        # already normalised (:41-42)
    scores = ops.cosine_scores_f64(predicted.contiguous(), gallery)
    top, _ = ops.topk_from_scores(scores, min(50, gallery.shape[0]))
    tgt = torch.tensor([name2idx[str(t)] for t in target_names], dtype=torch.int32, device=device)
    hit = top == tgt[:, None]
    n = len(target_names)
    return (hit[:, :10].any(dim=1).sum().item() / n * 100, hit[:, :50].any(dim=1).sum().item() / n * 100)


def generate_cirr_val_predictions(model, relative_val_dataset, index_names: List[str], index_features: torch.Tensor,
                                  device=torch.device("cuda")):
    """-> (predicted_features, reference_names, target_names, group_members)   tgcir/validate.py:197-244"""
    name2idx = _name_index(index_names)
    preds, refs, tgts, groups = [], [], [], []
    for batch in _batches(relative_val_dataset):
        ridx = torch.tensor([name2idx[str(b[0])] for b in batch], dtype=torch.int64, device=index_features.device)
        pred = _fuse(model, index_features[ridx].to(device), [b[2] for b in batch])
        preds.append(_normalized(pred, device))                        # F.normalize of tgcir/validate.py:238
        refs.extend(b[0] for b in batch)
        tgts.extend(b[1] for b in batch)
        groups.extend(list(b[3]) for b in batch)
    return torch.vstack(preds), refs, tgts, groups


def compute_cirr_val_metrics(relative_val_dataset, model, index_features: torch.Tensor,
                             index_features_normed_pooled: torch.Tensor, index_names: List[str],
                             device=torch.device("cuda")):
    """blip4cir/validate.py:133-195 -> (Rs@1, Rs@2, Rs@3, R@1, R@5, R@10, R@50) in percent."""
    predicted, reference_names, target_names, group_members = generate_cirr_val_predictions(
        model, relative_val_dataset, index_names, index_features, device)
    name2idx = _name_index(index_names)
    gallery = index_features_normed_pooled.to(device, torch.float32).contiguous()
    n = len(target_names)
    ref_idx = torch.tensor([name2idx[str(r)] for r in reference_names], dtype=torch.int32, device=device)
    tgt_idx = torch.tensor([name2idx[str(t)] for t in target_names], dtype=torch.int64, device=device)
    scores = ops.cosine_scores_f64(predicted.contiguous(), gallery)
    top, _ = ops.topk_from_scores(scores, min(50, gallery.shape[0] - 1), exclude=ref_idx)    # reference removed (:166-170)
    hit = top == tgt_idx[:, None].to(torch.int32)
    recalls = [hit[:, :k].any(dim=1).sum().item() / n * 100 for k in (1, 5, 10, 50)]
    gm = torch.tensor([[name2idx[str(m)] for m in g] for g in group_members], dtype=torch.int64, device=device)
    gs = torch.gather(scores, 1, gm)
    ts = torch.gather(scores, 1, tgt_idx[:, None])
    is_ref = gm == ref_idx[:, None].to(torch.int64)
    ahead = ((gs > ts) | ((gs == ts) & (gm < tgt_idx[:, None]))) & ~is_ref
    rank = ahead.sum(dim=1)
    assert bool((gm == tgt_idx[:, None]).any(dim=1).all()), "every target must be one of its group members"
    groups = [(rank < k).sum().item() / n * 100 for k in (1, 2, 3)]
    return tuple(groups + recalls)
