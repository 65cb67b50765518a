"""Oracle: the negative-type ablation losses of clip4cir/models_negtype.py (CPU fp32; test infrastructure only).

Restates models_negtype.py:53-85 (text_neg_loss, refer_neg_loss, query_neg_loss), :94-128 (forward: the four in-batch
InfoNCE terms and their neg_type bit mask) and :130-134 (infonce_loss) at the FEATURE level: inputs are the raw reference
image features, the text features and the raw target image features (what the towers return); the towers themselves are
pinned elsewhere (oracle/clip_text.py, oracle/clip_vision.py).

    q_i      = normalize(refer_i + text_i)                      t_i = normalize(target_i)
    target   = CE_i( <q_i, t_j> / tau over j, label i )         bit 4   (the ordinary in-batch loss)
    query    = CE_i( <t_i, q_j> / tau over j, label i )         bit 8
    text     = CE_i( <normalize(refer_i + text_j), t_i> / tau over j, label i )    bit 2   (negative TEXTS for query i)
    refer    = CE_i( <normalize(refer_j + text_i), t_i> / tau over j, label i )    bit 1   (negative REFERENCE images)
    loss     = mean of the selected terms (every CE is a mean over i)
"""
import torch
import torch.nn.functional as F


def terms(refer, text, target, tau):
    """-> dict of the four scalar losses (torch tensors, differentiable)."""
    B = refer.shape[0]
    lab = torch.arange(B)
    t = F.normalize(target, dim=-1)
    q = F.normalize(refer + text, dim=-1)
    out = {"target": F.cross_entropy(q @ t.t() / tau, lab),                      # models_negtype.py:104
           "query": F.cross_entropy(t @ q.t() / tau, lab)}                       # :107
    # :53-66: for query i the B candidates are normalize(refer_i + text_j), scored against target i only
    qt = F.normalize(refer[:, None, :] + text[None, :, :], dim=-1)               # [i, j, D]
    out["text"] = F.cross_entropy(torch.einsum("ijd,id->ij", qt, t) / tau, lab)
    # :68-80: for text i the candidates are normalize(refer_j + text_i)
    qr = F.normalize(refer[None, :, :] + text[:, None, :], dim=-1)               # [i, j, D]
    out["refer"] = F.cross_entropy(torch.einsum("ijd,id->ij", qr, t) / tau, lab)
    return out


def loss(refer, text, target, tau, neg_type):
    """models_negtype.py:108-127: bits 8 / 4 / 2 / 1 select query / target / text / refer; the mean of the selected terms."""
    tm = terms(refer, text, target, tau)
    sel = [tm[k] for bit, k in ((8, "query"), (4, "target"), (2, "text"), (1, "refer")) if neg_type & bit]
    if not sel:
        raise ValueError("neg_type selects no loss term")
    return sum(sel) / len(sel)
