#!/usr/bin/env python
"""Stage-2 hot-path benchmark: triplets/sec of the SPN4CIR second-stage step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = the loop body of clip4cir/train_negplus.py:107-123 on config 2 of BASELINE.json
(CLIP ViT-L/14 text tower, B = 256 per GPU, 77-token captions, 40 000 x 768 static negative bank,
tau 0.02, AdamW lr 2e-5): token ids already on the device -> text tower fwd -> combiner + L2-norm
-> bank InfoNCE -> backward -> (gradient all-reduce) -> AdamW -> bf16 weight refresh.
Synthetic data, seeded random-init weights (no datasets / checkpoints offline).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel's achieved FLOP/s, timed live with HIP events on its stream
  cpu_baseline - the CPU oracle (a port of the reference's fp32 CPU path) timed on the host cores
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# The HSA runtime reads these when it is initialised (the first torch.cuda call), RCCL at init_process_group: they
# have to be in the environment before either.  dmabuf IPC is the only mode the host driver supports (without it
# RCCL's buffer exchange fails with hipIpcGetMemHandle: invalid argument); rendezvous goes over loopback.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E spec
FLOP_PER_TRIPLET = 40.0e9     # 3 x 13.30 G (tower) + 2 x 2MD + projection (the REFERENCE's arithmetic)
# What the path EXECUTES per triplet: since round 5 the last block's out-projection + MLP (90.8 + 726.7 MFLOP forward per 77-row
# sequence) run on the pooled EOT row only - forward, data gradient and weight gradient lose 76 / 77 of that
# (spn_text_cfg.pool; SPN_POOL_LAST=0 computes every row again)
POOLED_SAVING = 3 * (90.8e6 + 726.7e6) * 76.0 / 77.0


def executed_flop_per_triplet():
    from spn4cir_amd import _lib
    off = (_lib.env("SPN_POOL_LAST") or "").strip() == "0"
    return FLOP_PER_TRIPLET - (0.0 if off else POOLED_SAVING)

# attention at L = 77, head_dim 64 has 39 flop per byte of q/k/v/o traffic (ridge: 312): it is bounded by HBM, and is
# priced on its algorithmic bytes (fwd: read qkv, write o = 8 W B per token; bwd: read qkv, o, dO, write dqkv = 16 W B)
KERNELS = {0: ("gemm_nt_kernel", "mfma"), 1: ("gemm_tn (grouped weight gradients, no split-K, + tail reduce)", "mfma"),
           2: ("attention_small_fwd_kernel", "hbm"), 3: ("attention_small_bwd_kernel", "hbm"),
           4: ("bank forward pass (160-row-tile kernel at 128 <= B <= 256, keeps p^T for the backward pass)", "hbm"),
           5: ("bank backward pass (G^T scale + dq GEMM + split-k fold from the saved p^T at B >= 128)", "hbm")}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch-per-gpu", type=int, default=256)
    p.add_argument("--bank", type=int, default=40000)
    p.add_argument("--model", default="ViT-L/14")
    p.add_argument("--tau", type=float, default=0.02)
    p.add_argument("--bank-mode", default="auto", choices=["auto", "sharded", "replicated"],
                   help="N > 1: how the bank loss is parallelised (DESIGN.md section 6).  auto = replicated below 10^6 bank rows "
                        "(no data-path collective: at M = 40 000 the sharded variant puts three latency-bound collectives on the "
                        "critical path to save ~40 us of bank kernel), sharded above (north_star's all-gather of queries)")
    p.add_argument("--no-alt-bank-mode", action="store_true",
                   help="N > 1: skip the short extra measurements reported as bank_mode_alt (the OTHER bank mode), strong (B_global = "
                        "256 split over the ranks) and grad_comm_bf16 (bf16 gradient buckets)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-prof", action="store_true")
    p.add_argument("--prof-every", type=int, default=17,
                   help="HIP-event pairs go around every n-th launch of the dominant kernel inside the timed region "
                        "(a stride coprime to the 392 gemm_nt launches of a step - 2^3 x 7^2 - samples every shape evenly; 1 = all launches)")
    p.add_argument("--kernel-pass-steps", type=int, default=3,
                   help="extra steps AFTER the timed region with every kernel class timed, for the kernels[] breakdown")
    p.add_argument("--no-packed", action="store_true", help="skip the extra (not headline) packed-EOT measurement")
    p.add_argument("--no-recall", action="store_true", help="skip the Recall@10 / top-K identity block")
    p.add_argument("--recall-queries", type=int, default=2000,
                   help="queries of the synthetic retrieval check (SURVEY 8d: 2 000; the CPU oracle encodes every one of "
                        "them in fp32, ~0.05 s each on 32 threads)")
    p.add_argument("--blip-images", type=int, default=30000,
                   help="images in the blip_config4 block's device-resident token bank (config 4: 30 000 = 26.6 / 35.4 GB bf16)")
    p.add_argument("--no-extra-configs", action="store_true",
                   help="skip the blip_config4 / fp8_config5 blocks (BASELINE configs 4 and 5 on one GPU, after the timed region)")
    p.add_argument("--recall-gallery", type=int, default=6000)
    p.add_argument("--cpu-batch", type=int, default=16)
    p.add_argument("--cpu-steps", type=int, default=3)
    p.add_argument("--cpu-threads", type=int, default=32)
    p.add_argument("--cpu-budget-s", type=float, default=20.0)
    return p.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (never exec: this
    process must not touch the GPU - no torch.cuda call has happened yet), relay its output with rank 0's JSON line
    last, and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.pop("MASTER_PORT", None)          # the default set above belongs to single-process runs
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's peer mappings need it on this driver
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = p.stdout.splitlines()
    result = [l for l in lines if l.startswith('{"metric"')]
    for l in lines:
        if not l.startswith('{"metric"'):
            print(l)
    for l in result[-1:]:
        print(l, flush=True)
    return p.returncode


def cpu_baseline(args, sd, target, refer):
    """The oracle (a CPU port of the reference's fp32 path: text fwd -> bank_large_step -> autograd
    backward -> torch AdamW as train_negplus.py:77-83 configures it) on the host cores."""
    from oracle import bank_loss, clip_text
    from spn4cir_amd import synthetic
    B = args.cpu_batch
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # torch's intra-op pool stops scaling (and thrashes) far below the core count of a big host
    threads = max(1, min(avail, args.cpu_threads))
    torch.set_num_threads(threads)
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW([{"params": list(params.values()), "lr": 2e-5, "betas": (0.9, 0.999), "eps": 1e-7}])
    ids = synthetic.token_ids(B, seed=1)
    ridx, labels = synthetic.triplet_indices(B, target.shape[0], seed=4)

    def step():
        opt.zero_grad()
        feats = clip_text.encode_text(params, ids)
        loss = bank_loss.bank_large_step(refer, ridx, feats, target, labels, args.tau)
        loss.backward()
        opt.step()
        return loss.item()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while n < args.cpu_steps and (n == 0 or time.perf_counter() - t0 < args.cpu_budget_s) and warm < 4 * args.cpu_budget_s:
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n if n else warm      # a pathological host: report the warm-up step
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(B / dt, 3), "unit": "triplets/sec", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "sample": f"{max(n, 1)} steps of B={B} at the config-2 shape (ViT-L/14 text tower, M={target.shape[0]}, "
                      f"D={target.shape[1]}, fp32, torch CPU kernels, {threads} threads of {avail} available cores), "
                      f"after 1 warm-up step",
            "ms_per_step": round(dt * 1e3, 1)}


def cpu_baseline_config1(args):
    """BASELINE config 1 on the host cores (clip4cir/train.py --wo_bank -> clip4cir/models.py:151-167): CLIP ViT-B/32
    (text 512 x 12 x 8 heads, vision 768 x 12 x 12 heads, patch 32, D = 512), B = 4, in-batch negatives, both towers
    trainable, AdamW lr 2e-5 - the oracle's restatement (oracle.clip_text / clip_vision / bank_loss.inbatch_step) with
    torch autograd; the reference additionally recomputes the towers under torch.utils.checkpoint (+1 forward)."""
    from oracle import bank_loss, clip_text, clip_vision
    from spn4cir_amd import synthetic
    B = 4
    threads = torch.get_num_threads()
    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS["ViT-B/32"]
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    sd.update(clip_vision.synthetic_vision_state_dict(768, 12, 32, 224, D, seed=5))
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW([{"params": list(params.values()), "lr": 2e-5, "betas": (0.9, 0.999), "eps": 1e-7}])
    ids = synthetic.token_ids(B, seed=1)
    g = torch.Generator().manual_seed(0)
    ref_img, tgt_img = torch.randn(B, 3, 224, 224, generator=g), torch.randn(B, 3, 224, 224, generator=g)

    def step():
        opt.zero_grad()
        loss = bank_loss.inbatch_step(clip_vision.encode_image(params, ref_img), clip_text.encode_text(params, ids),
                                      clip_vision.encode_image(params, tgt_img), args.tau)
        loss.backward()
        opt.step()
        return loss.item()

    step()
    n, t0 = 0, time.perf_counter()
    while n < 3 and (n == 0 or time.perf_counter() - t0 < args.cpu_budget_s / 2):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B / dt, 3), "unit": "triplets/sec", "cores": threads, "kind": "port", "ms_per_step": round(dt * 1e3, 1),
            "sample": f"{n} steps of B={B}, CLIP ViT-B/32 (both towers trainable, 224x224 images, in-batch negatives, fp32, "
                      f"autograd without the reference's checkpoint recompute), after 1 warm-up step"}


def blip_config4_block(args, dev):
    """BASELINE config 4 on ONE GPU (blip4cir/train.py:110-129 -> models.py:95-121): BERT-base fusion encoder (12 x 768,
    cross-attention over 577 image tokens), B = 128, 32-token captions, 30 000 x 256 bank, tau 0.03 (learnable), AdamW,
    for the reference's encoder width (768) and BASELINE's ViT-L (1024).  Not the headline."""
    from spn4cir_amd.fusion import BlipStage2Trainer, FusionEncoder
    out = {}
    B, L, S, M, images = 128, 32, 577, 30000, args.blip_images
    for E in (768, 1024):
        g = torch.Generator().manual_seed(0)
        enc = FusionEncoder(768, 12, 12, 3072, E, 256, 30524, 512, dev)
        with torch.no_grad():
            for k, v in enc.named_views().items():
                if k.endswith("LayerNorm.weight"):
                    v.fill_(1.0)
                elif v.dim() >= 2:
                    v.copy_((torch.randn(v.shape, generator=g) * 0.02).to(dev))
        enc.mark_stale()
        ids = torch.randint(1000, 30522, (B, L), generator=g, dtype=torch.int32)
        ids[:, 0] = 30523
        lens = torch.randint(6, L + 1, (B,), generator=g)
        mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int32)
        ids, mask = (ids * mask).to(dev), mask.to(dev)
        # the reference-token bank of ALL 30 000 images, bf16, resident on the device (26.6 GB at enc_width 768,
        # 35.4 GB at 1024; the reference keeps it fp32 in host RAM and uploads 227 MB per step); filled in chunks (set-up, untimed)
        ref_bank = torch.empty(images, S, E, dtype=torch.bfloat16, device=dev)
        dgen = torch.Generator(device=dev).manual_seed(5)
        for s0 in range(0, images, 1000):
            n = min(1000, images - s0)
            ref_bank[s0:s0 + n].copy_(torch.randn(n, S, E, generator=dgen, device=dev, dtype=torch.float32))
        ridx = torch.randint(0, images, (B,), generator=g).to(dev)
        labels = torch.randint(0, M, (B,), generator=g).to(dev)
        tr = BlipStage2Trainer(enc, tau=0.03, lr=5e-6, bank_mode="replicated")
        tr.set_bank(torch.nn.functional.normalize(torch.randn(M, 256, generator=g)))
        tr.set_token_bank(ref_bank)
        T, TS, W, I = B * L, B * S, 768, 3072
        fwd = 12 * (2 * T * W * 3 * W + 2 * T * W * W * 2 + 2 * TS * E * 2 * W + 2 * T * W * I * 2
                    + 4 * B * 12 * L * L * 64 + 4 * B * 12 * L * S * 64)
        # the reference's arithmetic (med.py:178-181 K/V projections of the 577 tokens per layer; dense B x L text rows): forward +
        # weight gradients + data gradients, EXCEPT the K/V projections' data gradient - the image tokens are detached
        # (models.py:97-100).  Rounds 3-4 quoted 3 x fwd, which counted that product too.
        alg = 3 * fwd - 12 * 2 * TS * E * 2 * W
        res = {}
        # "dense": attention mask on the device (never inspected: the padded B x L rows, as the reference computes them);
        # "packed": the tokenizer's host mask -> only the unmasked text rows (spn_fusion_cfg.T; same features and gradients)
        for rows, m in (("dense", mask), ("packed", mask.cpu())):
            for _ in range(3):
                loss = tr.step(ids, m, None, labels, token_idx=ridx)
            torch.cuda.synchronize()
            n, t0 = 10, time.perf_counter()
            for _ in range(n):
                loss = tr.step(ids, m, None, labels, token_idx=ridx)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            res[rows] = {"value": round(B / dt, 1), "unit": "triplets/sec", "ms_per_step": round(dt * 1e3, 3),
                         "loss_last": round(float(loss.item()), 5)}
            if rows == "dense":
                # priced at the REFERENCE's arithmetic (K/V-projection form, every text row through all 12 layers); the path executes
                # less: absorbed cross-attention (204 instead of 369 GFLOP per layer) and a pooled last layer
                res[rows].update({"reference_arith_tflops": round(alg / dt / 1e12, 1),
                                  "frac_of_bf16_peak_reference_arith": round(alg / dt / 1e12 / PEAK_BF16_TFLOPS, 4)})
            else:
                res[rows]["text_rows_live"] = f"{int(lens.sum())} of {B * L}"
        out[f"enc_width_{E}"] = res
        del tr, enc, ref_bank
        torch.cuda.empty_cache()
    out["workload"] = (f"blip4cir stage-2 step: BERT-base fusion (12 x 768, cross-attention over {S} image tokens), B={B}, "
                       f"L={L}, bank {M}x256, tau 0.03 learnable, AdamW; reference tokens gathered by the library "
                       f"(spn_fusion_fwd_bank) from a device-resident [{images}, {S}, E] bf16 token bank "
                       f"({images * S * 768 * 2 / 1e9:.1f} / {images * S * 1024 * 2 / 1e9:.1f} GB); caption lengths uniform in 6..{L}; cross-attention "
                       "in the absorbed form (csrc/xattn.hip); 10 steps after 3 warm-up; 1 GPU")
    return out


def jpeg_decode_block(dev):
    """SURVEY 8f-3 (image decode in front of the frozen image tower): batched GPU decode of FashionIQ-sized baseline JPEGs
    (spn_jpeg_decode_batch) beside Pillow on one host core; pixels compared with Pillow's on the spot."""
    import io
    import numpy as np
    from PIL import Image
    from spn4cir_amd import jpeg
    rng = np.random.default_rng(0)
    files = []
    for _ in range(8):
        h, w = 600, 400
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.zeros((h, w, 3))
        for _k in range(10):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(30, 200)
            img += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))[..., None] * rng.uniform(0, 255, 3)
        img = img / img.max() * 255 + rng.normal(0, 3, (h, w, 3))
        buf = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=90)
        files.append(buf.getvalue())
    t0 = time.perf_counter()
    ref = [np.asarray(Image.open(io.BytesIO(f)).convert("RGB")) for f in files]
    pil_rate = len(files) / (time.perf_counter() - t0)
    out = {"file_kb": round(sum(len(f) for f in files) / len(files) / 1024, 1), "image": "400 x 600, 4:2:0, quality 90",
           "pillow_one_host_core_images_per_s": round(pil_rate, 1)}
    jpeg.decode_batch(files, dev)
    torch.cuda.synchronize()
    for n in (256, 1024):
        fl = (files * (n // len(files)))[:n]
        t0 = time.perf_counter()
        dec, fb = jpeg.decode_batch(fl, dev)
        torch.cuda.synchronize()
        out[f"gpu_images_per_s_batch_{n}"] = round(n / (time.perf_counter() - t0), 1)
    out["bit_identical_to_pillow"] = bool(not fb and all(np.array_equal(dec[i].cpu().numpy(), ref[i % len(files)]) for i in range(16)))
    out["note"] = ("whole call: host marker parsing + upload + Huffman (a lane per file) + IDCT + upsampling / colour; used by the bank "
                   "builders and extract_index_features in 1 024-item chunks")
    return out


def _fp8_min_b():
    from spn4cir_amd import ops
    return ops.fp8_image_min_b()


def fp8_config5_block(args, sd, model_cls, dev):
    """BASELINE config 5's bank on ONE GPU: M = 100 000 rows x 768 stored e4m3 + per-row scale, the fp8-MFMA similarity pass,
    beside the same step on the bf16 bank, at the per-GPU batches of the 8-GPU run (32 = strong scaling, 256 = weak)."""
    from spn4cir_amd import synthetic
    from spn4cir_amd.trainer import Stage2Trainer
    M = 100000
    D = sd["text_projection"].shape[1]
    target, refer = synthetic.banks(M, D, seed=2)
    model = model_cls(sd, tau=args.tau, device=dev, plus=True)
    lib = __import__("spn4cir_amd._lib", fromlist=["lib"]).lib()
    out = {}
    for B in (32, 256):
        ids = synthetic.token_ids(B, seed=1).to(dev)
        ridx, labels = [t.to(dev) for t in synthetic.triplet_indices(B, M, seed=4)]
        ent = {}
        for dt_name in ("fp8", "bf16"):
            model.tower.load_clip_state_dict(sd)          # both bank formats start from the same weights
            tr = Stage2Trainer(model, lr=2e-5)
            tr.set_banks(refer, target, bank_dtype=dt_name)
            first = None
            for _ in range(3):
                loss = tr.step(ids, ridx, labels)
                if first is None:
                    first = loss.clone()
            torch.cuda.synchronize()
            n, t0 = 8, time.perf_counter()
            for _ in range(n):
                loss = tr.step(ids, ridx, labels)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            lib.spn_prof_enable(256)
            lib.spn_prof_select(0x30, 1)
            for _ in range(4):
                tr.step(ids, ridx, labels)
            torch.cuda.synchronize()
            lib.spn_prof_disable()
            pair = 0.0
            for kid in (4, 5):
                ms, work, nl = C.c_double(), C.c_double(), C.c_int()
                lib.spn_prof_collect(kid, C.byref(ms), C.byref(work), C.byref(nl))
                pair += ms.value / max(1, nl.value) * 1e3
            lib.spn_prof_reset()
            from spn4cir_amd import ops as _ops
            operand = _ops.bank_operand_kind(tr._bank, B)          # what the kernels READ: e4m3 bytes, bf16, or the kept bf16 image
            nbytes = M * D * (1 if operand == "e4m3" else 2)
            ent[dt_name] = {"triplets_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "bank_pair_us": round(pair, 1),
                            "operand": operand,
                            "bank_GBps_one_read": round(nbytes / (pair * 1e-6) / 1e9, 1) if pair > 0 else None,
                            "loss_first_step": round(float(first.item()), 5), "loss_last": round(float(loss.item()), 5)}
            del tr
        # SURVEY 8d's gate for config 5: |delta loss| <= 1e-2 between the two bank formats on the same weights (first step)
        ent["loss_abs_diff_first_step"] = round(abs(ent["fp8"]["loss_first_step"] - ent["bf16"]["loss_first_step"]), 6)
        out[f"B{B}"] = ent
    out["workload"] = (f"config-2 step (ViT-L/14 text tower) over a {M} x {D} bank stored e4m3 + fp32 row scale vs bf16; "
                       f"bank_pair_us = HIP-event time of the bank forward + backward kernels; `operand` = what those kernels read "
                       f"(from {_fp8_min_b()} queries per call an e4m3 bank is served through its kept bf16 image - "
                       f"SPN_FP8_IMAGE_MIN_B - so the B256 'fp8' entry is NOT an fp8-MFMA pass); bank_GBps_one_read prices the "
                       f"operand's bytes; 8 steps after 3 warm-up; 1 GPU")
    del model
    torch.cuda.empty_cache()
    return out


def recall_block(args, sd, model, dev):
    """BASELINE's metric is "triplets/sec + Recall@10": the validate.py:19-51 surface on a synthetic FashionIQ-shaped
    retrieval task (6 000 x 768 gallery), scored against the CPU oracle (oracle/clip_text.py fp32 tower +
    oracle/recall.py fp64 ranking).  Gallery = seeded random unit rows; a query = normalize(gallery[ref] + text(caption))
    (validate.py:84-95); each query's target row is PLANTED at normalize(q_oracle + 0.222 * noise), i.e. at the cosine
    (~0.16) of the best random rows, so that Recall@10 is neither 0 nor 1 and moves with the feature error.  Reported
    for the validation default (fp32-exact tower) and for the bf16 training tower: Recall@10/50 of the HIP path and of the
    oracle, and the fraction of queries whose top-10 / top-50 index SETS (reference row removed, validate.py:39) are
    identical to the oracle's (north_star: identical top-K index sets)."""
    import numpy as np
    from oracle import clip_text, recall
    from spn4cir_amd import ops, synthetic
    nq, ng = args.recall_queries, args.recall_gallery
    D = model.tower.embed_dim
    if nq <= 0 or nq > ng // 2:
        return None
    g = torch.Generator().manual_seed(12)
    gallery = torch.nn.functional.normalize(torch.randn(ng, D, generator=g))
    ref_idx = torch.randint(0, ng // 2, (nq,), generator=g)
    tgt_idx = ng // 2 + torch.randperm(ng - ng // 2, generator=g)[:nq]
    ids = synthetic.token_ids(nq, seed=11)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), args.cpu_threads)))
    t0 = time.perf_counter()
    with torch.no_grad():
        t_cpu = torch.cat([clip_text.encode_text(sd, ids[i:i + 64]) for i in range(0, nq, 64)])
    cpu_s = time.perf_counter() - t0
    q_cpu = torch.nn.functional.normalize(gallery[ref_idx] + t_cpu)
    gallery[tgt_idx] = torch.nn.functional.normalize(q_cpu + 0.222 * torch.randn(nq, D, generator=g))
    order, scores_cpu = recall.ranked_indices(q_cpu.numpy(), gallery.numpy())
    ref_np, tgt_np = ref_idx.numpy(), tgt_idx.numpy()

    def top_sets(order_rows):                         # drop the reference row, keep the first 50 (validate.py:39-45)
        out = []
        for i in range(nq):
            row = order_rows[i]
            out.append(row[row != ref_np[i]][:51])          # 50 + the first row outside the top-50 (boundary gap below)
        return np.stack(out)

    top_cpu = top_sets(order[:, :52])
    gal_d, ref_d = gallery.to(dev), ref_idx.to(dev)
    excl = ref_idx.to(dev, torch.int32)

    def gpu_top(text_feats):
        q, _, _ = ops.combine_l2norm_fwd(gal_d, ref_d, text_feats.float().contiguous())
        gn, _, _ = ops.combine_l2norm_fwd(None, None, gal_d)
        idx, _ = ops.topk_from_scores(ops.cosine_scores_f64(q, gn), 50, exclude=excl)
        return idx.cpu().numpy()

    def report(top):
        r = {}
        for K in (10, 50):
            r[f"recall_at_{K}"] = round(float((top[:, :K] == tgt_np[:, None]).any(1).mean() * 100), 3)
            r[f"topk_identical_frac_{K}"] = round(float(np.mean([set(top[i, :K]) == set(top_cpu[i, :K]) for i in range(nq)])), 5)
        return r

    ids_d = ids.to(dev)
    model.tower.load_clip_state_dict(sd)              # the timed steps moved the weights: back to the oracle's
    with torch.no_grad():
        t0 = time.perf_counter()
        t_exact = torch.cat([model.tower.forward_exact(ids_d[i:i + 256].contiguous()) for i in range(0, nq, 256)])
        torch.cuda.synchronize()
        exact_s = time.perf_counter() - t0
        t_bf16 = torch.cat([model.tower.forward(ids_d[i:i + 256].contiguous()).clone() for i in range(0, nq, 256)])
    top_ex = gpu_top(t_exact)
    ex, bf = report(top_ex), report(gpu_top(t_bf16))
    cpu = report(top_cpu)
    # a top-K set that differs from the oracle's: how close were the oracle's own scores at the boundary?  (the exact tower's
    # features equal the oracle's to ~1e-7 relative - accumulation order - so a set can only flip where rank K and K + 1 tie to
    # that precision)
    ties = {}
    for K in (10, 50):
        gaps = []
        for i in range(nq):
            if set(top_ex[i, :K]) != set(top_cpu[i, :K]):
                sc = scores_cpu[i, top_cpu[i, :K + 1]]
                gaps.append(float(sc[K - 1] - sc[K]))
        ties[str(K)] = {"queries": len(gaps), "max_oracle_score_gap_at_boundary": max(gaps) if gaps else 0.0}
    cos = torch.nn.functional.cosine_similarity(t_exact.cpu().double(), t_cpu.double(), dim=-1)
    cosb = torch.nn.functional.cosine_similarity(t_bf16.cpu().double(), t_cpu.double(), dim=-1)
    return {"recall_at_10": ex["recall_at_10"], "recall_at_50": ex["recall_at_50"],
            "recall_at_10_oracle": cpu["recall_at_10"], "recall_at_50_oracle": cpu["recall_at_50"],
            "topk_identical_frac": {"10": ex["topk_identical_frac_10"], "50": ex["topk_identical_frac_50"]},
            "topk_mismatch": ties,
            "tower": "fp32-exact (the default of the validate.py surface)",
            "text_feature_max_1_minus_cos": float((1 - cos).max()),
            "bf16_training_tower": {"recall_at_10": bf["recall_at_10"], "recall_at_50": bf["recall_at_50"],
                                    "topk_identical_frac": {"10": bf["topk_identical_frac_10"], "50": bf["topk_identical_frac_50"]},
                                    "text_feature_max_1_minus_cos": float((1 - cosb).max())},
            "queries": nq, "gallery_rows": ng, "embed_dim": D,
            "sample": f"{nq} queries (SURVEY 8d: 2 000; --recall-queries); oracle encode {cpu_s:.1f} s on the host cores, "
                      f"exact tower {exact_s * 1e3:.0f} ms on the GPU"}


def _build_info():
    """How __graft_entry__.build() produced the library this run loaded: compiled by that call or reused (spn4cir_amd/build_info.json)."""
    try:
        with open(os.path.join(ROOT, "spn4cir_amd", "build_info.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {"mode": "unknown (no build_info.json: the library was not built through __graft_entry__.build())"}


def collect_kernels(lib, n_steps, rows, W):
    """kernels[] entries from the library's HIP-event records (spn_prof_*), after n_steps steps with every class timed."""
    ks = []
    for kid, (name, bound) in KERNELS.items():
        ms, work, n = C.c_double(), C.c_double(), C.c_int()
        lib.spn_prof_collect(kid, C.byref(ms), C.byref(work), C.byref(n))
        if not n.value:
            continue
        w = work.value
        if kid in (2, 3):     # the profiler counts flops for attention; price it on bytes (see KERNELS)
            w = n.value * float(rows) * W * 2.0 * (4 if kid == 2 else 8)
        rate = w / (ms.value * 1e-3)
        ent = {"kernel": name, "bound": bound, "launches_per_step": n.value // max(1, n_steps),
               "avg_us": round(ms.value / n.value * 1e3, 1), "ms_per_step": round(ms.value / max(1, n_steps), 3)}
        if bound == "mfma":
            ent.update(achieved=round(rate / 1e12, 1), unit="TFLOP/s", frac=round(rate / 1e12 / PEAK_BF16_TFLOPS, 4))
        else:
            ent.update(achieved=round(rate / 1e9, 1), unit="GB/s", frac=round(rate / 1e9 / PEAK_HBM_GBS, 4))
        ks.append(ent)
    return ks


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            raise SystemExit(self_launch(args.gpus))     # one process per GPU under torch.distributed.run, as a child
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product")
    # SPN_BENCH_SHARE_GPU=1 (debugging aid for 1-GPU boxes): every rank uses device 0 and the process group runs over gloo -
    # the N-rank control flow of this script (sharding, bank modes, touched-row exchange, max-over-ranks timing) end to end
    # without N GPUs; the number it prints is meaningless
    share = os.environ.get("SPN_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    # SPN_DP_FORCE_COLLECTIVES=1 (1-GPU debugging aid): run the N-GPU code path - RCCL init, sharded bank, bucketed
    # gradient all-reduce - with a 1-rank group
    force_dp = os.environ.get("SPN_DP_FORCE_COLLECTIVES") == "1"
    if world > 1 or force_dp:
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        dist.barrier()
        C.CDLL(None).fflush(None)          # emit RCCL's start-up banner now, not after the result line

    from spn4cir_amd import _lib, synthetic
    from spn4cir_amd.models import CIRPlus
    from spn4cir_amd.trainer import Stage2Trainer

    W, layers, heads, D = synthetic.CLIP_TEXT_CONFIGS[args.model]
    sd = synthetic.text_state_dict(W, layers, D, seed=0)
    model = CIRPlus(sd, tau=args.tau, device=dev, plus=True)
    target, refer = synthetic.banks(args.bank, D, seed=2)
    if args.bank_mode == "auto":
        args.bank_mode = "replicated" if args.bank < 1000000 else "sharded"
    # pack=False: `value` is the dense 77-token computation the reference performs (the drop-in's DEFAULT is the packed mode,
    # reported separately below as `packed_eot`)
    trainer = Stage2Trainer(model, lr=2e-5, group=group, bank_mode=args.bank_mode, pack=False)
    trainer.set_banks(refer, target)

    B = args.batch_per_gpu
    B_global = B * world
    ids_all = synthetic.token_ids(B_global, seed=1)
    ridx_all, lab_all = synthetic.triplet_indices(B_global, args.bank, seed=4)
    sl = slice(rank * B, (rank + 1) * B)
    ids, ridx, labels = ids_all[sl].to(dev), ridx_all[sl].to(dev), lab_all[sl].to(dev)
    # N > 1: the host copy of this rank's ids lets the trainer exchange the token-embedding gradient as touched rows
    # (SparseRowReducer) instead of a dense 152 MB all-reduce; a training loop has them on the host anyway (tokenizer output)
    ids_host = ids_all[sl].contiguous() if (world > 1 or force_dp) and os.environ.get("SPN_DENSE_EMBED_ALLREDUCE") != "1" else None

    def barrier():
        if world > 1 or force_dp:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = trainer.step(ids, ridx, labels, ids_host=ids_host)
    lib = _lib.lib()
    prof = not args.no_prof
    # Live roofline measurement: HIP-event pairs around the launches of the DOMINANT kernel (gemm_nt, checked
    # below against the all-kernel pass) inside the timed region, around every --prof-every-th launch (default 17:
    # ~460 samples in 20 steps, every GEMM shape hit evenly).  An event pair keeps a launch from overlapping its
    # neighbours: bracketing every kernel class costs 7 % of the step, every gemm_nt launch 2.5 %, every 5th 0.7 %, every 17th 0.2 %.
    # The full per-class breakdown comes from a few extra steps after the timed region.
    DOM = 0
    if prof:
        lib.spn_prof_enable(max(64, 200 * (args.steps + args.kernel_pass_steps)))
        lib.spn_prof_select(1 << DOM, max(1, args.prof_every))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(ids, ridx, labels, ids_host=ids_host)
    barrier()
    dt = time.perf_counter() - t0
    dom_rec = None
    if prof:
        ms, work, n = C.c_double(), C.c_double(), C.c_int()
        lib.spn_prof_collect(DOM, C.byref(ms), C.byref(work), C.byref(n))
        dom_rec = (ms.value, work.value, n.value)
        lib.spn_prof_reset()
        lib.spn_prof_select(0xFFFFFFFF, 1)
        for _ in range(args.kernel_pass_steps):
            trainer.step(ids, ridx, labels, ids_host=ids_host)
        torch.cuda.synchronize()
        lib.spn_prof_disable()
        dense_raw = {}
        for kid in KERNELS:                   # collected now: the packed pass below reuses the recorder
            ms, work, n = C.c_double(), C.c_double(), C.c_int()
            lib.spn_prof_collect(kid, C.byref(ms), C.byref(work), C.byref(n))
            dense_raw[kid] = (ms.value, work.value, n.value)
        lib.spn_prof_reset()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    loss_val = float(loss.item())

    # Extra, NOT the headline: the same step with the text tower in packed mode (rows after each caption's EOT
    # token are dead under the causal mask, so they are not computed).  `value` above is the dense 77-token
    # computation the reference performs; this only reports what the optional mode buys on these caption lengths.
    packed = None
    if not args.no_packed:
        # as a training loop runs it: the ids also on the host every step (they come from the tokenizer there), the trainer
        # derives the prefix sums and uploads them through its pinned staging buffer - no stream synchronisation
        ids_host_p = ids_all[sl]
        _, total = trainer.tower.cu_seqlens(ids_host_p)
        ids_p = ids[:, :trainer.tower.live_length(ids_host_p)].contiguous()     # padding columns only beyond the longest caption
        trainer.pack = True
        for _ in range(max(2, args.warmup)):
            lp = trainer.step(ids_p, ridx, labels, ids_host=ids_host_p)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lp = trainer.step(ids_p, ridx, labels, ids_host=ids_host_p)
        barrier()
        dtp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
        packed = {"value": round(B_global * args.steps / dtp.item(), 1), "unit": "triplets/sec",
                  "ms_per_step": round(dtp.item() / args.steps * 1e3, 3), "live_rows_rank0": total,
                  "dense_rows_per_rank": B * ids.shape[1], "loss_last": round(float(lp.item()), 5),
                  "note": "TextTower packed mode = the DEFAULT of CIRPlus (pack_eot) and Stage2Trainer (pack) since round 5: "
                          "bit-identical features; per-step host prefix sums + pinned upload included; not the headline value"}
        if prof:
            # the same per-class breakdown for the packed step (rocprofv3 table: profiles/r03_packed_kernel_stats.txt);
            # every rank runs the steps (they contain collectives), rank 0 reports its records
            lib.spn_prof_enable(max(64, 200 * args.kernel_pass_steps))
            lib.spn_prof_select(0xFFFFFFFF, 1)
            for _ in range(args.kernel_pass_steps):
                trainer.step(ids_p, ridx, labels, ids_host=ids_host_p)
            torch.cuda.synchronize()
            lib.spn_prof_disable()
            packed["kernels"] = collect_kernels(lib, args.kernel_pass_steps, total, W)
            lib.spn_prof_reset()

    trainer.pack = False
    # N > 1: the other bank mode, measured briefly with the same barrier / max-over-ranks protocol (not the headline)
    extras_dead = [False]

    def guarded(fn):
        """An extra measurement must never cost the headline: a failure is reported in the JSON line instead of raised.  N > 1: a
        rank that failed alone has left its peers inside a collective or will skip the next one - the ranks agree on an error flag
        after every extra (MAX all-reduce), and once any rank failed, no further extra (each of them enters collectives) is started
        on any rank."""
        if extras_dead[0]:
            return {"error": "skipped: an earlier extra measurement failed on some rank"}
        err = None
        try:
            res = fn()
        except Exception as exc:          # noqa: BLE001
            err = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if world > 1:
            try:
                flag = torch.tensor([1.0 if err else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0:
                    extras_dead[0] = True
                    err = err or {"error": "another rank failed in this measurement"}
            except Exception as exc:      # noqa: BLE001  (the group itself is broken: nothing collective can follow)
                extras_dead[0] = True
                err = err or {"error": f"error-flag exchange failed: {exc}"[:300]}
        return err if err else res

    alt = None
    if (world > 1 or force_dp) and not args.no_alt_bank_mode:
        def measure_alt():
            other = "sharded" if args.bank_mode == "replicated" else "replicated"
            tr2 = Stage2Trainer(model, lr=2e-5, group=group, bank_mode=other, pack=False)
            tr2.set_banks(refer, target)
            for _ in range(max(2, args.warmup)):
                tr2.step(ids, ridx, labels, ids_host=ids_host)
            barrier()
            t0 = time.perf_counter()
            n_alt = max(5, args.steps // 2)
            for _ in range(n_alt):
                tr2.step(ids, ridx, labels, ids_host=ids_host)
            barrier()
            dta = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(dta, op=dist.ReduceOp.MAX)
            return {"bank_mode": other, "value": round(B_global * n_alt / dta.item(), 1), "unit": "triplets/sec",
                    "ms_per_step": round(dta.item() / n_alt * 1e3, 3), "steps": n_alt}
        alt = guarded(measure_alt)

    # N > 1: two more brief measurements with the same protocol, so that the first multi-GPU run settles what DESIGN.md section 6
    # only predicts - (a) STRONG scaling: config 3's B_global = 256 split over the ranks (256 / N per GPU; SURVEY 8d asks for
    # both regimes), (b) the gradient buckets exchanged as bf16 (all-to-all + fp32 rank-order sum + all-gather) at the weak shape
    strong, comm16, comm_direct, collectives, zero1, zero1_strong = None, None, None, None, None, None
    if (world > 1 or force_dp) and not args.no_alt_bank_mode:
        def brief(tr, ids_, ridx_, labels_, host_, bglob):
            for _ in range(max(2, args.warmup)):
                tr.step(ids_, ridx_, labels_, ids_host=host_)
            barrier()
            t0_ = time.perf_counter()
            n_ = max(5, args.steps // 2)
            for _ in range(n_):
                tr.step(ids_, ridx_, labels_, ids_host=host_)
            barrier()
            d_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(d_, op=dist.ReduceOp.MAX)
            return {"value": round(bglob * n_ / d_.item(), 1), "unit": "triplets/sec", "ms_per_step": round(d_.item() / n_ * 1e3, 3),
                    "steps": n_}
        def measure_strong(**kw):
            bs = 256 // world
            ids_s = synthetic.token_ids(256, seed=1)
            ridx_s, lab_s = synthetic.triplet_indices(256, args.bank, seed=4)
            sl_s = slice(rank * bs, (rank + 1) * bs)
            tr3 = Stage2Trainer(model, lr=2e-5, group=group, bank_mode=args.bank_mode, pack=False, **kw)
            tr3.set_banks(refer, target)
            host_s = ids_s[sl_s].contiguous() if ids_host is not None else None
            r = brief(tr3, ids_s[sl_s].to(dev), ridx_s[sl_s].to(dev), lab_s[sl_s].to(dev), host_s, 256)
            r.update(scaling="strong", global_batch=256, batch_per_gpu=bs, bank_mode=args.bank_mode)
            r.update(kw)
            return r

        def measure_comm(**kw):
            tr4 = Stage2Trainer(model, lr=2e-5, group=group, bank_mode=args.bank_mode, pack=False, **kw)
            tr4.set_banks(refer, target)
            r = brief(tr4, ids, ridx, labels, ids_host, B_global)
            r.update(kw)
            return r
        if 256 % world == 0 and 256 // world >= 8:
            strong = guarded(measure_strong)
        comm16 = guarded(lambda: measure_comm(grad_comm_dtype="bf16"))
        comm16["note"] = ("weak-scaling shape; dense gradient buckets cross the links as bf16 (all-to-all, fp32 sum in rank order, "
                          "all-gather); headline uses fp32 all-reduce")
        def measure_collectives():
            """achieved all-reduce bandwidth and the overlap of the gradient exchange with backward.  (1) the dense
            gradient buckets alone, in the trainer's own bucket pattern, nothing else on the device; (2) the weak-shape step
            with the bucket exchange switched off (same kernels, replicas drift - measurement only); overlap = the share of (1)
            that the headline step hides: 1 - (step - step_without_exchange) / exchange_alone."""
            tr6 = Stage2Trainer(model, lr=2e-5, group=group, bank_mode=args.bank_mode, pack=False)
            spans = tr6.tower.layer_spans()
            tok = tr6.tower.vocab * tr6.tower.width if ids_host is not None else 0        # rows exchanged sparsely in the step
            elems = sum(e - max(s_, tok) for s_, e in spans if e > tok)

            def exchange():
                for s_, e in spans:
                    tr6.reducer.on_span_ready(max(s_, tok), e) if e > max(s_, tok) else None
                tr6.reducer.finish()
            for _ in range(2):
                exchange()
            barrier()
            t0_ = time.perf_counter()
            n_ = 5
            for _ in range(n_):
                exchange()
            barrier()
            d_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(d_, op=dist.ReduceOp.MAX)
            t_ex = d_.item() / n_
            q_local = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
            q_all = torch.empty(B_global, D, dtype=torch.bfloat16, device=dev)
            for _ in range(3):
                dist.all_gather_into_tensor(q_all, q_local)
            barrier()
            t0_ = time.perf_counter()
            for _ in range(20):
                dist.all_gather_into_tensor(q_all, q_local)
            barrier()
            t_ag = (time.perf_counter() - t0_) / 20
            del tr6
            nocomm = measure_comm(grad_comm_algo="none")
            bytes_ = elems * 4
            out_ = {"dense_gradient_bytes": bytes_, "exchange_alone_ms": round(t_ex * 1e3, 3),
                    "allreduce_busbw_GBps": round(bytes_ * 2 * (world - 1) / max(world, 1) / t_ex / 1e9, 1) if world > 1 else None,
                    "allgather_queries_us": round(t_ag * 1e6, 1), "allgather_queries_bytes": B_global * D * 2,
                    "step_without_exchange_ms": nocomm.get("ms_per_step"), "headline_step_ms": round(dt * 1e3 / args.steps, 3),
                    "xgmi_links_per_gpu": 7,
                    "note": "busbw = bytes x 2 (G - 1) / G / time (the all-reduce convention); per link: busbw / 7 when every link "
                            "carries an equal share"}
            if nocomm.get("ms_per_step") and t_ex > 0:
                hidden = 1.0 - (out_["headline_step_ms"] - nocomm["ms_per_step"]) / (t_ex * 1e3)
                out_["overlap_fraction"] = round(max(0.0, min(1.0, hidden)), 3)
            return out_
        collectives = guarded(measure_collectives)
        # sharded optimizer step (ZeRO-1 shape, Stage2Trainer(optim="sharded")): the direct exchange's reduced chunk is updated by
        # its owner and the masters are all-gathered - beside the replicated update of the headline, at both scaling shapes
        zero1 = guarded(lambda: measure_comm(optim="sharded"))
        zero1["note"] = ("weak-scaling shape; AdamW runs on each rank's 1 / G slice of every bucket, the updated fp32 masters are "
                         "all-gathered instead of the reduced gradients (same link bytes as grad_comm_direct_fp32)")
        if 256 % world == 0 and 256 // world >= 8:
            zero1_strong = guarded(lambda: measure_strong(optim="sharded"))
        comm_direct = guarded(lambda: measure_comm(grad_comm_algo="direct"))
        comm_direct["note"] = ("weak-scaling shape; fp32 buckets through all-to-all + rank-order sum + all-gather (every xGMI link at "
                               "once) instead of RCCL's all-reduce")

    if rank == 0:
        per_kernel = {}
        if prof:
            for kid, (name, bound) in KERNELS.items():
                msv, workv, nv = dense_raw[kid]
                if nv:
                    w = workv
                    if kid in (2, 3):     # the profiler counts flops for attention; price it on bytes (see KERNELS)
                        w = nv * float(B * ids.shape[1]) * W * 2.0 * (4 if kid == 2 else 8)
                    per_kernel[kid] = dict(kernel=name, bound=bound, launches=nv, total_ms=msv,
                                           avg_us=msv / nv * 1e3, work=w)
        roof = None
        extra = {}
        pass_ms = None
        if prof and dom_rec and dom_rec[2]:
            dms, dwork, dn = dom_rec
            name, bound = KERNELS[DOM]
            ach = dwork / (dms * 1e-3) / 1e12
            # the events cover every prof_every-th launch: scale the sampled time to all launches for the share
            launches_total = dn * max(1, args.prof_every)
            traffic, traffic_src = None, None
            import glob
            tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
            tpath = tfiles[-1] if tfiles else ""          # the latest round's PMC passes
            if tpath and args.batch_per_gpu == 256 and args.model == "ViT-L/14":
                # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command
                # (bench.py cannot profile itself); FETCH_SIZE already doubled per the gfx950 correction
                with open(tpath) as f:
                    tj = json.load(f)
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/" + os.path.basename(tpath)
            roof = {"bound": bound, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    # the PMC passes cannot run inside this process: the figure is REPLAYED from the committed rocprofv3 passes of this
                    # same command (profiles/), not measured by this run
                    "traffic_replayed": traffic is not None,
                    "kernel": name, "launches": dn, "avg_us": round(dms / dn * 1e3, 1),
                    "share_of_step": round(dms / dn * launches_total / (dt * 1e3), 3),
                    "timed": f"HIP-event pairs around every {max(1, args.prof_every)}-th launch inside the timed region",
                    "clock_note": "peak is the 2.4 GHz dense bf16 figure; the matrix kernels run power-limited at 1.6-2.0 GHz "
                                  "on this part (rocm-smi samples: profiles/r02_gemm_power_clock.txt)"}
        if per_kernel:
            pass_ms = sum(k["total_ms"] for k in per_kernel.values())
            dom = max((k for k in per_kernel.values() if k["bound"] == "mfma"), key=lambda k: k["total_ms"])
            if roof is not None and dom["kernel"] != roof["kernel"]:
                roof["note"] = f"the all-kernel pass ranks {dom['kernel']} first"
            ks = []
            for k in per_kernel.values():
                rate = k["work"] / (k["total_ms"] * 1e-3)
                ent = {"kernel": k["kernel"], "bound": k["bound"], "launches_per_step": k["launches"] // max(1, args.kernel_pass_steps),
                       "avg_us": round(k["avg_us"], 1),
                       "ms_per_step": round(k["total_ms"] / max(1, args.kernel_pass_steps), 3)}
                if k["bound"] == "mfma":
                    ent.update(achieved=round(rate / 1e12, 1), unit="TFLOP/s", frac=round(rate / 1e12 / PEAK_BF16_TFLOPS, 4))
                else:
                    ent.update(achieved=round(rate / 1e9, 1), unit="GB/s", frac=round(rate / 1e9 / PEAK_HBM_GBS, 4))
                ks.append(ent)
            extra["kernels"] = ks
            extra["kernels_pass"] = (f"{args.kernel_pass_steps} extra steps AFTER the timed region with every kernel class "
                                     f"timed (that instrumentation slows a step by ~7 %, so it stays out of `value`)")
        tps = B_global * args.steps / dt
        out = {
            "metric": "triplets/sec", "value": round(tps, 1), "unit": "triplets/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"clip4cir {args.model} text tower + {args.bank}x{D} static negative bank "
                                   f"(train_negplus.py second stage), B={B}/GPU, 77-token captions, tau={args.tau}, "
                                   f"AdamW lr 2e-5, random-init weights",
                       "global_batch": B_global, "seq_len": 77, "bank_rows": args.bank, "embed_dim": D,
                       "parallelism": f"dp{world}" + (f"+bank-{args.bank_mode}" if world > 1 else "")},
            "loss_last": round(loss_val, 5),
            # primary: the flops the path EXECUTES (dense 77-row captions; the pooled last block runs on B rows)
            "step_model_tflops": round(tps * executed_flop_per_triplet() / 1e12 / world, 1),
            "step_frac_of_bf16_peak": round(tps * executed_flop_per_triplet() / 1e12 / world / PEAK_BF16_TFLOPS, 4),
            # the same step priced at the reference's arithmetic (every one of the 77 rows through all 12 blocks)
            "step_reference_arith_tflops": round(tps * FLOP_PER_TRIPLET / 1e12 / world, 1),
            "step_frac_of_bf16_peak_reference_arith": round(tps * FLOP_PER_TRIPLET / 1e12 / world / PEAK_BF16_TFLOPS, 4),
            "step_flop_note": f"executed = {executed_flop_per_triplet() / 1e9:.2f} GFLOP per triplet against the reference's "
                              f"{FLOP_PER_TRIPLET / 1e9:.1f}: the last block's out-projection / MLP (forward, data and weight "
                              "gradients) run on the B pooled EOT rows only (spn_text_cfg.pool; clip/model.py:352-356 reads one row "
                              "per caption); `roofline.achieved` counts the flops of the launches actually made",
            "roofline": roof,
            # every SPN_* variable the library saw when it was loaded (its A/B switches read that snapshot only): a stray
            # one that changes a kernel is visible next to the number
            "lib_config": dict(_lib.config_dump(), build=_build_info()),
        }
        out.update(extra)
        if packed:
            out["packed_eot"] = packed
        if alt:
            out["bank_mode_alt"] = alt
        if strong:
            out["strong"] = strong
        if comm16:
            out["grad_comm_bf16"] = comm16
        if comm_direct:
            out["grad_comm_direct_fp32"] = comm_direct
        if zero1:
            out["optim_sharded"] = zero1
        if zero1_strong:
            out["optim_sharded_strong"] = zero1_strong
        if collectives:
            out["collectives"] = collectives
        if world == 1 and not args.no_recall:     # checker legs run at N = 1 only: the other ranks would sit in the exit barrier
            rec = recall_block(args, sd, model, dev)
            if rec:
                out["recall_at_10"] = rec["recall_at_10"]
                out["topk_identical_frac"] = rec["topk_identical_frac"]
                out["recall"] = rec
        if world == 1 and not args.no_extra_configs:
            del trainer
            torch.cuda.empty_cache()

            def dropin_block():
                """What a maintainer who only swaps the import sees: the reference's OWN loop bodies
                (train_negplus.py:107-123, blip4cir/train.py:110-129) on the drop-in CIRPlus objects - caption strings tokenised
                on the host every step, autograd backward, GradScaler, a host read of the loss per step - with torch.optim.AdamW
                and with the one-line switch to spn4cir_amd.optim.AdamW.  tools/dropin_step.py, tools/dropin_blip_step.py."""
                import importlib.util
                res = {}
                for key, fname, kw in (("clip_config2", "dropin_step.py", {}), ("blip_config4", "dropin_blip_step.py", {"images": 2000})):
                    spec = importlib.util.spec_from_file_location("spn_" + key, os.path.join(ROOT, "tools", fname))
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    res[key] = [mod.run(fused_optim=f, steps=8, warmup=2, **kw) for f in (False, True)]
                    torch.cuda.empty_cache()
                res["note"] = ("strings in, loss read on the host every step (as the reference formats it), packed rows (the default for "
                               "host ids / masks); compare with packed_eot (CLIP) and blip_config4.*.packed (fused trainers, no host sync)")
                return res
            out["dropin_loop"] = guarded(dropin_block)
            torch.cuda.empty_cache()
            out["blip_config4"] = guarded(lambda: blip_config4_block(args, dev))       # extras never cost the headline line
            torch.cuda.empty_cache()
            out["fp8_config5"] = guarded(lambda: fp8_config5_block(args, sd, CIRPlus, dev))
            out["jpeg_decode"] = guarded(lambda: jpeg_decode_block(dev))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, sd, target, refer)
            out["cpu_baseline"]["config1"] = cpu_baseline_config1(args)
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL prints a version banner to the C stdout buffer: flush it first so that the JSON line is the LAST line
    try:
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
