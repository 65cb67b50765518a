"""Golden vectors for the BLIP-2 / Q-Former second-stage loss, captured by running the reference's own
`Blip2QformerCirAlignPrompt.forward_stage2` (blip24cir/lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:
226-268) in the build container (needs /root/reference).  Run in its own interpreter:

    python tests/golden/make_golden_blip2.py

The module's imports (lavis.common.registry, lavis.models.base_model, lavis.models.blip2_models.blip2,
lavis.models.blip_models.blip_outputs) pull in omegaconf / timm / fairscale / iopath, absent offline.  They are replaced
by import-only stand-ins: `registry.register_model(name)` returns the identity decorator (it only registers the class),
`Blip2Base` is an empty nn.Module subclass (base class, never instantiated here), every other symbol raises if called.
`forward_stage2` is then called UNBOUND on a plain namespace that carries exactly the attributes the method reads:

    query_tokens [1,32,H], device, max_txt_len, temp (learnable, nn.Parameter), text_proj_q (a real nn.Linear),
    tokenizer      -> returns fixed input_ids / attention_mask (the BERT vocabulary is not available offline),
    Qformer_query.bert(...) -> returns an object whose last_hidden_state is a fixed leaf tensor `hidden` [B, 32+L, H]
                              (the Q-Former is out of scope; its output is an INPUT of the pinned lines).

What runs as the reference wrote it is everything behind the Q-Former call (:247-268): text_proj_q + F.normalize of
position 32, the per-sample loop (matmul with target_feats.permute(0,2,1), max over the 32 token rows, / temp,
cross_entropy), the batch mean - and autograd through it.

blip2_stage2.npz: hidden [B,40,H], text_proj_q weight/bias, target_indexs, temp (target_feats [M,32,256] is regenerated
from its seed on both sides: cases.blip2_target_feats) ->
fusion_feats, loss_qtc, d loss / d fusion_feats (via a hook), d hidden[:,32,:], d temp, d text_proj_q.*;
two cases (M = 37 with ties between token rows, M = 1000)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from cases import BLIP2_CASES, blip2_target_feats  # noqa: E402
REF = "/root/reference/blip24cir/lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py"


def _never(name):
    def f(*a, **k):
        raise RuntimeError(f"import-only stub {name} was executed")
    return f


def load_module():
    names = ("lavis", "lavis.common", "lavis.common.registry", "lavis.models", "lavis.models.base_model",
             "lavis.models.blip2_models", "lavis.models.blip2_models.blip2", "lavis.models.blip_models",
             "lavis.models.blip_models.blip_outputs")
    mods = {n: types.ModuleType(n) for n in names}

    class _Registry:
        @staticmethod
        def register_model(name):
            return lambda cls: cls

    mods["lavis.common.registry"].registry = _Registry
    bm = mods["lavis.models.base_model"]
    bm.all_gather_with_grad, bm.concat_all_gather = _never("all_gather_with_grad"), _never("concat_all_gather")
    b2 = mods["lavis.models.blip2_models.blip2"]

    class Blip2Base(nn.Module):
        pass

    b2.Blip2Base, b2.compute_sim_matrix, b2.disabled_train = Blip2Base, _never("compute_sim_matrix"), _never("disabled_train")
    bo = mods["lavis.models.blip_models.blip_outputs"]
    bo.BlipOutput, bo.BlipOutputFeatures = _never("BlipOutput"), _never("BlipOutputFeatures")
    saved = {n: sys.modules.get(n) for n in mods}
    sys.modules.update(mods)
    try:
        spec = importlib.util.spec_from_file_location("ref_blip2_cir", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    return mod


class _Tokens:
    def __init__(self, ids, mask):
        self.input_ids, self.attention_mask = ids, mask

    def to(self, device):
        return self


def run_case(mod, tag, B, M, H, D, L, seed, ties):
    g = torch.Generator().manual_seed(seed)
    hidden = torch.randn(B, 32 + L, H, generator=g, requires_grad=True)
    proj = nn.Linear(H, D)
    with torch.no_grad():
        proj.weight.copy_(0.1 * torch.randn(D, H, generator=g))
        proj.bias.copy_(0.05 * torch.randn(D, generator=g))
    target_feats = blip2_target_feats(tag)     # regenerated from its seed by the tests (32 MB at M = 1000)
    target_indexs = torch.randint(0, M, (B,), generator=g)
    if ties:
        target_indexs[0], target_indexs[1] = 3, 7
    temp = nn.Parameter(torch.tensor(0.07))
    ids = torch.randint(1, 1000, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    grabbed = {}

    def bert(input_ids, query_embeds=None, attention_mask=None, return_dict=True):
        assert query_embeds is grabbed["fusion_hidden_states"] and attention_mask.shape == (B, 32 + L)
        return types.SimpleNamespace(last_hidden_state=hidden)

    me = types.SimpleNamespace(
        query_tokens=torch.zeros(1, 32, H), device=torch.device("cpu"), max_txt_len=L, temp=temp, text_proj_q=proj,
        tokenizer=lambda text, **kw: _Tokens(ids, mask), Qformer_query=types.SimpleNamespace(bert=bert))
    grabbed["fusion_hidden_states"] = torch.randn(B, 32, H, generator=g)
    # d loss / d fusion_feats: hook on the output of F.normalize inside the method (via text_proj_q's output graph)
    feats_grad = {}
    orig_normalize = mod.F.normalize

    def normalize_hook(x, *a, **k):
        y = orig_normalize(x, *a, **k)
        if y.requires_grad:
            y.retain_grad()
            feats_grad["y"] = y
        return y

    mod.F.normalize = normalize_hook
    try:
        out = mod.Blip2QformerCirAlignPrompt.forward_stage2(me, ["caption"] * B, target_feats,
                                                            grabbed["fusion_hidden_states"], target_indexs)
    finally:
        mod.F.normalize = orig_normalize
    loss = out["loss_qtc"]
    loss.backward()
    y = feats_grad["y"]
    return {"hidden": hidden.detach().numpy(), "proj_w": proj.weight.detach().numpy(), "proj_b": proj.bias.detach().numpy(),
            "target_indexs": target_indexs.numpy(), "temp": np.float32(temp.item()),
            "fusion_feats": y.detach().numpy(), "loss_qtc": np.float32(loss.item()), "d_fusion_feats": y.grad.numpy(),
            "d_hidden32": hidden.grad[:, 32, :].numpy(), "d_temp": np.float32(temp.grad.item()),
            "d_proj_w": proj.weight.grad.numpy(), "d_proj_b": proj.bias.grad.numpy()}


def main():
    mod = load_module()
    out = {}
    for tag, kw in BLIP2_CASES.items():
        for k, v in run_case(mod, tag, **kw).items():
            out[f"{tag}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "blip2_stage2.npz"), **out)
    print("wrote blip2_stage2.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "target_feats" not in k})


if __name__ == "__main__":
    main()
