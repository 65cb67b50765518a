"""GPU parity of the TG-CIR second-stage step vs vectors captured from tgcir/models.py on CPU
(tests/golden/make_golden_tgcir.py) and vs the oracle restatement (oracle/tgcir_head.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _model():
    from cases import TGCIR, tgcir_inputs, tgcir_weights
    from spn4cir_amd.tgcir_models import CIRPlus
    sd, head = tgcir_weights()
    ids, ref, bank, labels = tgcir_inputs()
    m = CIRPlus(sd, tau=TGCIR["TAU"], plus=True)
    m.load_head(head)
    m.refer_bank = ref
    m.target_bank = bank
    return m, sd, head, ids, ref, bank, labels


def test_tgcir_head_kernels_match_oracle():
    """Head alone in fp32-in / fp32-out terms: the oracle on the SAME token features (so only the two bf16 GEMMs
    differ), forward and every head gradient + the gradients handed back to the text tower."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import tgcir_head
    from spn4cir_amd.tgcir_models import TgcirHead
    g = torch.Generator().manual_seed(3)
    B, L, C = 5, 77, 512
    head = tgcir_head.synthetic_head(C, 8, 4, seed=11)
    tokens = torch.randn(B, L, C, generator=g)
    feats = torch.randn(B, C, generator=g)
    ref = torch.randn(B, 12, C, generator=g) * 0.5
    dpooled = torch.randn(B, C, generator=g)
    h = TgcirHead(C, 4, "cuda")
    h.load(head)
    tb = tokens.cuda().to(torch.bfloat16)
    pooled, mod = h.forward(feats.cuda(), tokens.cuda(), tb, ref.cuda())
    # oracle on the bf16-rounded tokens (what the text_fc GEMM sees)
    hd = {k: v.double().requires_grad_(True) for k, v in head.items()}
    tok_d = tb.cpu().double().requires_grad_(True)
    feats_d = feats.double().requires_grad_(True)
    mod_ref = tgcir_head.extract_text_fea(tok_d, feats_d, hd)
    x = torch.cat([ref.double(), mod_ref], dim=-1)
    hh = torch.relu(x @ hd["s_remain_map.0.weight"].t() + hd["s_remain_map.0.bias"])
    remain = torch.sigmoid(hh @ hd["s_remain_map.2.weight"].t() + hd["s_remain_map.2.bias"])
    pooled_ref = (remain * ref.double() + (1 - remain) * mod_ref).mean(dim=1)
    assert rel(mod, mod_ref) < 5e-3
    assert rel(pooled, pooled_ref) < 5e-3
    (pooled_ref * dpooled.double()).sum().backward()
    dfeats, dtokens = h.backward(dpooled.cuda())
    assert rel(dfeats, feats_d.grad) < 2e-2
    assert rel(dtokens, tok_d.grad) < 2e-2
    gv = h.named_views(h.grads)
    for k in tgcir_head.HEAD_KEYS:
        # text_fc / TokenLearner see bf16 GEMM operands (dz, tokens): the training path's per-parameter gate (DESIGN.md
        # section 3); s_remain_map runs on the fp32-exact GEMMs end to end
        assert rel(gv[k], hd[k].grad.reshape(gv[k].shape)) < (1e-2 if k.startswith("s_remain_map") else 5e-2), k


def test_text_tower_token_output_and_backward():
    """spn_text_fwd_tokens / spn_text_bwd_tokens vs the oracle tower: ln_final of every position, both gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cases import tgcir_inputs, tgcir_weights
    from oracle import tgcir_head
    from spn4cir_amd.text_tower import TextTower, text_cfg_from_state_dict
    sd, _ = tgcir_weights()
    ids = tgcir_inputs()[0]
    c = text_cfg_from_state_dict(sd)
    t = TextTower(c["width"], c["layers"], c["heads"], c["embed_dim"], c["vocab"], c["ctx"], "cuda")
    t.load_clip_state_dict(sd)
    feats, tokens, tokens_b = t.forward_tokens(ids.cuda())
    sdd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tok_ref, feats_ref = tgcir_head.text_tokens(sdd, ids)
    assert rel(tokens, tok_ref) < 1e-2 and rel(feats, feats_ref) < 1e-2
    assert torch.equal(tokens_b.float(), tokens.to(torch.bfloat16).float())
    g = torch.Generator().manual_seed(9)
    dtok = torch.randn(tok_ref.shape, generator=g) * 0.1
    dfe = torch.randn(feats_ref.shape, generator=g)
    ((tok_ref * dtok).sum() + (feats_ref * dfe).sum()).backward()
    grads = t.named_views(t.backward_tokens(dfe.cuda(), dtok.cuda()))
    for k, v in grads.items():
        assert rel(v, sdd[k].grad) < 5e-2, k           # the training path's per-parameter gate


def test_tgcir_step_matches_reference(golden_dir):
    """CIRPlus.forward -> backward against the reference's own loss / query / mod tokens / gradients."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cases import tgcir_grad_check
    m, sd, head, ids, ref, bank, labels = _model()
    z = np.load(os.path.join(golden_dir, "tgcir_step.npz"))
    B = ids.shape[0]
    out = m.forward(ids, None, labels, torch.arange(B))
    assert set(out) == {"bank_loss"} and out["bank_loss"].dim() == 0
    out["bank_loss"].backward()
    assert abs(out["bank_loss"].item() - float(z["loss"])) < 2e-2 * max(1.0, abs(float(z["loss"])))
    q = m.img_txt_fusion(ref, ids)
    cos = torch.nn.functional.cosine_similarity(q.cpu().double(), torch.from_numpy(z["q"]).double(), dim=-1)
    assert (1 - cos).max() < 1e-3                                       # north_star gate on the embeddings
    assert rel(m.extract_text_fea(ids), torch.from_numpy(z["mod_token"])) < 1e-2
    for k in ("text_fc.weight", "text_fc.bias", "tokenlearn_text.weight", "tokenlearn_text.bias", "masks_text.weight",
              "s_remain_map.0.weight", "s_remain_map.0.bias", "s_remain_map.2.weight", "s_remain_map.2.bias"):
        # s_remain_map[0]: its gradient is a heavily cancelling sum over tokens behind a ReLU mask - in the fp32 oracle
        # itself a 0.3 % relative perturbation of the text features (what the bf16 tower introduces) moves it by 5-6 %
        # while every other parameter moves < 1 %; the head kernels alone reproduce it to 1e-2 (test above)
        tgcir_grad_check(z, k, m._params[k].grad, 1e-1 if k.startswith("s_remain_map.0") else 5e-2)
    for k in sd:
        tgcir_grad_check(z, "clip." + k, m._params["clip." + k].grad, 5e-2)
    # state-dict names follow the reference's modules
    names = dict(m.named_parameters())
    assert "backbone.clip.ln_final.weight" in names and "backbone.text_fc.weight" in names and "s_remain_map.0.weight" in names
