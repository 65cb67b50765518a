"""Recall@K validation surface of the reference on the MI355X kernels.

Mirrors clip4cir/validate.py: `compute_fiq_val_metrics` (:19-51), `generate_fiq_val_predictions`
(:54-98), `compute_cirr_val_metrics` (:111-156), `generate_cirr_val_predictions` (:159-213) - same
signatures, same return values (percentages as python floats).

What changed underneath: the reference builds the full N_q x N_g distance matrix in fp32 and
argsorts every row; here the query side is combiner + L2-normalise in one kernel, scores are
accumulated in fp64 (so the ranking does not depend on summation order)
and only the top-50 (reference image excluded in-kernel) is selected.  Datasets are duck-typed:
anything indexable that yields the reference's tuples.
"""
from typing import List, Tuple

import numpy as np
import torch

from . import ops


class _Slice:
    """Contiguous [begin, end) view of an indexable dataset (one rank's share of the validation queries)."""

    def __init__(self, ds, begin, end):
        self.ds, self.begin, self.end = ds, begin, end

    def __len__(self):
        return self.end - self.begin

    def __getitem__(self, i):
        return self.ds[self.begin + i]


def _shard(dataset, distributed, group):
    """validation shards the QUERY set across ranks, the gallery is replicated, metrics are summed counts."""
    import torch.distributed as dist
    if not distributed or not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dataset, None
    from .distributed import shard_range
    b, e = shard_range(len(dataset), dist.get_world_size(group), dist.get_rank(group))
    return _Slice(dataset, b, e), group


def _reduce_counts(counts, n, device, sharded, group):
    """[hits...] and the local query count -> percentages over the GLOBAL query count (identical on every rank)."""
    t = torch.tensor(list(counts) + [n], dtype=torch.float64, device=device)
    if sharded:
        import torch.distributed as dist
        dist.all_reduce(t, group=group)
    return [float(x) / float(t[-1]) * 100 for x in t[:-1]]


def _batches(dataset, bs=32):
    n = len(dataset)
    for s in range(0, n, bs):
        yield [dataset[i] for i in range(s, min(n, s + bs))]


def _name_index(index_names):
    return {str(n): i for i, n in enumerate(index_names)}


def _predict(model, captions, ref_idx, index_features):
    """normalize(combining_function(index_features[ref], encode_text(captions)))  (validate.py:84-95)

    Under torch.no_grad() as in the reference (validate.py:83): that is what selects CIRPlus(exact_eval=True)'s fp32
    tower, and a bf16 training forward here would overwrite the activations of a pending backward."""
    with torch.no_grad():
        text = model.encode_text(captions)
        fn = getattr(model, "combining_function", None)
        if fn is None or getattr(getattr(fn, "__func__", fn), "_spn_fused_sum", False):
            # the built-in element-wise sum (models_negplus.py:48-50): gather + add + normalise in ONE launch
            q, _, _ = ops.combine_l2norm_fwd(index_features, ref_idx, text.float().contiguous())
            return q
        # any other Combiner a user put on the model (validate.py:92,206 call the attribute): run it, then F.normalize
        mixed = fn(index_features[ref_idx], text)
        q, _, _ = ops.combine_l2norm_fwd(None, None, mixed.to(index_features.device, torch.float32).contiguous())
    return q


def generate_fiq_val_predictions(model, relative_val_dataset, index_names: List[str], index_features: torch.Tensor,
                                 device=torch.device("cuda")):
    """-> (predicted_features [N, D] L2-normalised, target_names, refer_names)   validate.py:54-98"""
    name2idx = _name_index(index_names)
    feats = index_features.to(device, torch.float32).contiguous()
    preds, target_names, refer_names = [], [], []
    for batch in _batches(relative_val_dataset):
        refs = [b[0] for b in batch]
        tgts = [b[1] for b in batch]
        # deterministic caption join, validate.py:73-79 (no capitalisation in clip4cir)
        caps = [f"{b[2][0].strip('.?, ')} and {b[2][1].strip('.?, ')}" for b in batch]
        ridx = torch.tensor([name2idx[str(r)] for r in refs], dtype=torch.int64, device=device)
        preds.append(_predict(model, caps, ridx, feats))
        target_names.extend(tgts)
        refer_names.extend(refs)
    return torch.vstack(preds), target_names, refer_names


def _scores_and_topk(predicted, index_features, exclude_idx, K):
    gal, _, _ = ops.combine_l2norm_fwd(None, None, index_features)      # F.normalize(index_features), validate.py:28
    scores = ops.cosine_scores_f64(predicted, gal)                      # 1 - distance, validate.py:31
    K = min(K, index_features.shape[0] - 1)
    idx, _ = ops.topk_from_scores(scores, K, exclude=exclude_idx)       # argsort + reference removal, :32,39
    return scores, idx


def compute_fiq_val_metrics(relative_val_dataset, model, index_features: torch.Tensor, index_names: List[str],
                            device=torch.device("cuda"), distributed=False, group=None) -> Tuple[float, float]:
    """validate.py:19-51 -> (recall@10, recall@50) in percent.  distributed=True: every rank scores its share of the
    queries against the whole gallery and the hit counts are all-reduced."""
    relative_val_dataset, grp = _shard(relative_val_dataset, distributed, group)
    sharded = isinstance(relative_val_dataset, _Slice)
    predicted, target_names, refer_names = generate_fiq_val_predictions(model, relative_val_dataset, index_names,
                                                                        index_features, device)
    name2idx = _name_index(index_names)
    feats = index_features.to(device, torch.float32).contiguous()
    ref_idx = torch.tensor([name2idx[str(r)] for r in refer_names], dtype=torch.int32, device=device)
    tgt_idx = torch.tensor([name2idx[str(t)] for t in target_names], dtype=torch.int32, device=device)
    _, top = _scores_and_topk(predicted, feats, ref_idx, 50)
    hit = top == tgt_idx[:, None]
    n = len(target_names)
    r10, r50 = _reduce_counts([hit[:, :10].any(dim=1).sum().item(), hit[:, :50].any(dim=1).sum().item()], n, device,
                              sharded, grp)
    return r10, r50


def generate_cirr_val_predictions(model, relative_val_dataset, index_names: List[str], index_features: torch.Tensor,
                                  device=torch.device("cuda")):
    """-> (predicted_features, reference_names, target_names, group_members)   validate.py:159-213"""
    name2idx = _name_index(index_names)
    feats = index_features.to(device, torch.float32).contiguous()
    preds, refs_all, tgts_all, groups = [], [], [], []
    for batch in _batches(relative_val_dataset):
        refs = [b[0] for b in batch]
        caps = [b[2] for b in batch]
        ridx = torch.tensor([name2idx[str(r)] for r in refs], dtype=torch.int64, device=device)
        preds.append(_predict(model, caps, ridx, feats))
        refs_all.extend(refs)
        tgts_all.extend(b[1] for b in batch)
        groups.extend(list(b[3]) for b in batch)
    return torch.vstack(preds), refs_all, tgts_all, groups


def compute_cirr_val_metrics(relative_val_dataset, model, index_features: torch.Tensor, index_names: List[str],
                             device=torch.device("cuda"), distributed=False, group=None):
    """validate.py:111-156 -> (Rs@1, Rs@2, Rs@3, R@1, R@5, R@10, R@50) in percent (distributed: see the FashionIQ one)."""
    relative_val_dataset, grp = _shard(relative_val_dataset, distributed, group)
    sharded = isinstance(relative_val_dataset, _Slice)
    predicted, reference_names, target_names, group_members = generate_cirr_val_predictions(
        model, relative_val_dataset, index_names, index_features, device)
    if index_features.dim() > 2:                                        # validate.py:120-121
        index_features = index_features.mean(dim=1)
    name2idx = _name_index(index_names)
    feats = index_features.to(device, torch.float32).contiguous()
    n = len(target_names)
    ref_idx = torch.tensor([name2idx[str(r)] for r in reference_names], dtype=torch.int32, device=device)
    tgt_idx = torch.tensor([name2idx[str(t)] for t in target_names], dtype=torch.int64, device=device)
    scores, top = _scores_and_topk(predicted, feats, ref_idx, 50)
    hit = top == tgt_idx[:, None].to(torch.int32)
    recall_counts = [hit[:, :k].any(dim=1).sum().item() for k in (1, 5, 10, 50)]
    # subset metric (validate.py:139-142): rank of the target among its group members, reference removed
    gm = torch.tensor([[name2idx[str(m)] for m in g] for g in group_members], dtype=torch.int64, device=device)
    gs = torch.gather(scores, 1, gm)                                    # [n, G] fp64
    ts = torch.gather(scores, 1, tgt_idx[:, None])                      # [n, 1]
    is_ref = gm == ref_idx[:, None].to(torch.int64)
    ahead = ((gs > ts) | ((gs == ts) & (gm < tgt_idx[:, None]))) & ~is_ref
    rank = ahead.sum(dim=1)
    in_group = (gm == tgt_idx[:, None]).any(dim=1)
    assert bool(in_group.all()), "every target must be one of its group members (validate.py:145)"
    group_counts = [(rank < k).sum().item() for k in (1, 2, 3)]
    return tuple(_reduce_counts(group_counts + recall_counts, n, device, sharded, grp))


def synthetic_recall_at_k(model, n_gallery=6000, n_query=2000, seed=7, device=torch.device("cuda")):
    """FashionIQ-shaped synthetic retrieval (no datasets offline): gallery = random features, each
    query's text feature points from its reference towards its target.  Returns (R@10, R@50)."""
    g = torch.Generator().manual_seed(seed)
    D = model.output_dim
    gallery = torch.randn(n_gallery, D, generator=g)
    ref = torch.randint(0, n_gallery, (n_query,), generator=g)
    tgt = (ref + 1 + torch.randint(0, n_gallery - 1, (n_query,), generator=g)) % n_gallery
    text = 0.6 * torch.randn(n_query, D, generator=g) + 0.9 * gallery[tgt] * torch.rand(n_query, 1, generator=g) \
        - 0.5 * gallery[ref]
    feats = gallery.to(device)
    q, _, _ = ops.combine_l2norm_fwd(feats, ref.to(device), text.to(device).contiguous())
    _, top = _scores_and_topk(q, feats, ref.to(device, torch.int32), 50)
    hit = top == tgt.to(device, torch.int32)[:, None]
    return (hit[:, :10].any(dim=1).float().mean().item() * 100, hit[:, :50].any(dim=1).float().mean().item() * 100)
