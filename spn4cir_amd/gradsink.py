"""nn.Parameter.grad over the towers' flat gradient buffers, with autograd's ACCUMULATE semantics.

The towers' backward passes OVERWRITE one flat fp32 buffer per tower (text_tower.backward & co).  The drop-in
`CIRPlus` objects expose slices of that buffer as `p.grad`, so a caller's optimizer (the reference's
`optim.AdamW(model.parameters())`, train_negplus.py:77-84) reads the gradients in place.  autograd, however,
ADDS into an existing `.grad`: two `backward()` calls without `zero_grad()` (gradient accumulation), or the
torch 1.13 default `zero_grad(set_to_none=False)` that keeps the tensors, must give old + new.  When `p.grad`
is the alias of the flat buffer the old value is destroyed by the tower's backward itself, so it has to be
saved BEFORE that call:

    snap = gradsink.snapshot(params, tower.grads, tower.named_views, prefix)
    flat = tower.backward(...)
    gradsink.publish(params, flat, tower.named_views, snap, prefix)
"""
import torch


def _aliases(grad, flat):
    return (grad is not None and grad.device == flat.device
            and grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr())


def _views(named_views, flat):
    """named_views(flat), memoised on the tower for its OWN persistent gradient buffer: building the ~150-330 views of a tower is
    ~1-1.5 ms of host time per call, paid twice per step (snapshot + publish) in front of / behind the backward launches of a
    loop that synchronises on the loss every step (the reference's does, train_negplus.py:114).  Other buffers (a snapshot's
    clone) are not cached: the cache would keep their memory alive."""
    owner = getattr(named_views, "__self__", None)
    if owner is None or getattr(owner, "grads", None) is not flat:
        return named_views(flat)
    key = (flat.data_ptr(), flat.numel())
    cached = owner.__dict__.get("_gradsink_views")
    if cached is None or cached[0] != key:
        cached = owner.__dict__["_gradsink_views"] = (key, named_views(flat))
    return cached[1]


def snapshot(params, flat, named_views, prefix=""):
    """Copy of `flat` if any parameter's .grad currently lives in it (its old gradient is about to be overwritten)."""
    for key in _views(named_views, flat):
        if _aliases(params[prefix + key].grad, flat):
            return flat.clone()
    return None


def publish(params, flat, named_views, snap, prefix=""):
    """After the tower's backward filled `flat`: p.grad = new (alias, when there was none) or old + new."""
    views = _views(named_views, flat)
    aliased = [k for k in views if _aliases(params[prefix + k].grad, flat)]
    if aliased:
        if snap is None:
            raise RuntimeError("gradsink.publish: a .grad aliases the flat buffer but no snapshot was taken before backward")
        if len(aliased) == len(views):
            flat.add_(snap)                       # every slice accumulates: one launch over the whole buffer
        else:
            old = named_views(snap)
            for k in aliased:
                views[k].add_(old[k])
    done = set(aliased)
    for k, view in views.items():
        if k in done:
            continue
        p = params[prefix + k]
        p.grad = view if p.grad is None else p.grad + view
