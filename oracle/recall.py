"""Oracle: Recall@K metrics (numpy).

Restates clip4cir/validate.py:19-51 (compute_fiq_val_metrics, after the predictions have
been generated) and :111-156 (compute_cirr_val_metrics).  Inputs here are already the
L2-normalised predicted query features, the raw gallery features and the name lists; the
feature-generation half (validate.py:54-98, :159-213) is model code and lives in the
product's host logic.

Scores are accumulated in float64 so that the ranking is the ranking of the exact
products of the fp32 inputs (independent of summation order); the reference uses an fp32
matmul whose order depends on the BLAS in use.  Ties are broken by the lower gallery index.
"""
import numpy as np


def _normalize(x, eps=1e-12):
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(-1, keepdims=True)).astype(np.float32)
    return x / np.maximum(n, eps)


def ranked_indices(pred, gallery):
    """distances = 1 - pred @ normalize(gallery)^T, ascending (validate.py:28-32)."""
    g = _normalize(gallery)
    scores = np.asarray(pred, dtype=np.float32).astype(np.float64) @ g.astype(np.float64).T
    return np.argsort(-scores, axis=-1, kind="stable"), scores


def fiq_recall(pred, gallery, index_names, target_names, refer_names):
    """validate.py:34-49: drop the reference from each ranking, then top-10 / top-50 membership."""
    order, _ = ranked_indices(pred, gallery)
    names = np.array(index_names)
    n = len(target_names)
    r10 = r50 = 0
    for i in range(n):
        row = names[order[i]]
        row = row[row != refer_names[i]]
        if target_names[i] in row[:10]:
            r10 += 1
            r50 += 1
        elif target_names[i] in row[:50]:
            r50 += 1
    return r10 / n * 100, r50 / n * 100


def cirr_recall(pred, gallery, index_names, reference_names, target_names, group_members):
    """validate.py:123-156.  Returns (Rs@1, Rs@2, Rs@3, R@1, R@5, R@10, R@50)."""
    order, _ = ranked_indices(pred, gallery)
    names = np.array(index_names)
    n = len(target_names)
    sorted_names = names[order]
    keep = sorted_names != np.array(reference_names).reshape(n, 1)
    sorted_names = sorted_names[keep].reshape(n, -1)                       # reference removed
    labels = sorted_names == np.array(target_names).reshape(n, 1)
    gm = np.array(group_members)
    group_mask = (sorted_names[..., None] == gm[:, None, :]).sum(-1).astype(bool)
    group_labels = labels[group_mask].reshape(n, -1)
    assert (labels.sum(-1) == 1).all()
    assert (group_labels.sum(-1) == 1).all()
    out = [group_labels[:, :k].sum() / n * 100 for k in (1, 2, 3)]
    out += [labels[:, :k].sum() / n * 100 for k in (1, 5, 10, 50)]
    return tuple(float(x) for x in out)
