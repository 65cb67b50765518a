"""End-to-end top-K identity report (SURVEY 7g level ii): query embeddings from the bf16 HIP text tower vs the fp32 CPU
oracle, both scored exactly (fp64) against the same gallery; reports the fraction of queries whose top-K index SETS are
identical and the score margin at rank K.  Kernel-level identity (same fp32 embeddings in -> same sets out) is a hard
test (tests/test_kernels_gpu.py::test_topk_identical_sets); this is the softer end-to-end number.

    python tools/recall_identity.py [--queries 64] [--gallery 6000]
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--gallery", type=int, default=6000)
    ap.add_argument("--model", default="ViT-B/32")
    a = ap.parse_args()
    from oracle import clip_text, recall
    from spn4cir_amd import ops, synthetic
    from spn4cir_amd.text_tower import TextTower
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS[a.model]
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    ids = synthetic.token_ids(a.queries, seed=1)
    g = torch.Generator().manual_seed(9)
    gallery = torch.nn.functional.normalize(torch.randn(a.gallery, D, generator=g))
    ref_feats = torch.randn(a.queries, D, generator=g)
    tower = TextTower(W, layers, heads, D, 49408, 77, "cuda")
    tower.load_clip_state_dict(sd)
    t_gpu = tower.forward(ids.cuda()).cpu()
    t_exact = tower.forward_exact(ids.cuda()).cpu()
    with torch.no_grad():
        t_cpu = clip_text.encode_text(sd, ids)
    out = {"model": a.model, "queries": a.queries, "gallery": a.gallery,
           "text_feature_1_minus_cos_max": float((1 - torch.nn.functional.cosine_similarity(t_gpu.double(), t_cpu.double())).max())}
    for name, gal, tq in (("isotropic random gallery, bf16 tower", gallery, t_gpu),
                          ("isotropic random gallery, fp32-exact tower (forward_exact)", gallery, t_exact)):
        pg = torch.nn.functional.normalize(ref_feats + tq)
        pc = torch.nn.functional.normalize(ref_feats + t_cpu)
        og, sg = recall.ranked_indices(pg.numpy(), gal.numpy())
        oc, sc = recall.ranked_indices(pc.numpy(), gal.numpy())
        res = {}
        for K in (1, 10, 50):
            same = [set(og[i, :K]) == set(oc[i, :K]) for i in range(a.queries)]
            srt = -np.sort(-sc, axis=1)
            res[f"top{K}_set_identity_rate"] = float(np.mean(same))
            res[f"median_margin_at_rank_{K}"] = float(np.median(srt[:, K - 1] - srt[:, K]))
        res["max_abs_score_difference"] = float(np.abs(sg - sc).max())
        out[name] = res
    # the device scorer on identical inputs is exact: same sets as the oracle
    idx, _ = ops.topk_from_scores(ops.cosine_scores_f64(pc.cuda().float(), gallery.cuda()), 50)
    out["device_scorer_identical_on_same_embeddings"] = bool((idx.cpu().numpy() == oc[:, :50]).all())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
