"""TG-CIR second stage behind the reference's own class protocol.

    from spn4cir_amd.tgcir_models import CIRPlus            # instead of tgcir/models.py
    loss = model.forward(text, indexs, target_indexs, refer_indexs)['bank_loss']; loss.backward()

Reference: tgcir/models.py - Backbone.extract_text_fea (:127-151), CIRPlus.img_txt_fusion (:198-205),
bank_large_step / infonce_loss (:272-296).  What is trainable in the second stage (load_ckpt(is_origin=True),
:210-221): the CLIP text tower, text_fc, tokenlearn_text, masks_text and the fusion MLPs; the image side is frozen
and enters only through the token bank `refer_bank` [N, 12, 512] and the pooled `target_bank` [M, 512].

Compute: spn_text_fwd_tokens / spn_text_bwd_tokens (text tower with ln_final of every position), bf16 MFMA GEMMs
for text_fc and s_remain_map[0], the spn_tg_* kernels (csrc/tgcir.hip) for TokenLearner / gating / pooling, and the
bank InfoNCE kernels.  The frozen image side (extract_img_fea / img_embed, :84-125,183-196: CLIP ViT tokens -> fc ->
TokenLearner + masked global tokens) runs on spn_vision_fwd's token output, a bf16 GEMM and the same TokenLearner
kernel; it serves the bank builders (:223-270), inference only.
"""
import os

import torch
from torch import nn

from . import gradsink, ops
from .preprocess import gpu_decode_scope, realize_items, stack_images
from ._lib import check, lib
from .ops import _p, _stream
from .text_tower import TextTower, text_cfg_from_state_dict
from .vision_tower import VisionTower, vision_cfg_from_state_dict

S_LOCAL = 8

# flat head layout: (state-dict style key, shape builder)
_HEAD = [("text_fc.weight", lambda C, S, G: (C, C)), ("text_fc.bias", lambda C, S, G: (C,)),
         ("tokenlearn_text.weight", lambda C, S, G: (S, C)), ("tokenlearn_text.bias", lambda C, S, G: (S,)),
         ("masks_text.weight", lambda C, S, G: (G, C)),
         ("s_remain_map.0.weight", lambda C, S, G: (C, 2 * C)), ("s_remain_map.0.bias", lambda C, S, G: (C,)),
         ("s_remain_map.2.weight", lambda C, S, G: (1, C)), ("s_remain_map.2.bias", lambda C, S, G: (1,))]
# every span starts on a 16-byte boundary (vector loads / stores in the kernels)


class TgcirHead:
    """Parameters (one flat fp32 buffer + a flat gradient buffer) and the forward / backward launch chains of the
    TG-CIR query producer that follows the text tower."""

    def __init__(self, C=512, G=4, device="cuda"):
        self.C, self.S, self.G, self.NT = C, S_LOCAL, G, G + S_LOCAL
        self.device = torch.device(device)
        self._spans, off = [], 0
        for key, shp in _HEAD:
            shape = shp(C, self.S, G)
            n = 1
            for s in shape:
                n *= s
            self._spans.append((key, off, shape))
            off += (n + 3) // 4 * 4
        self.n_params = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self._stale = True
        self._st = None

    def named_views(self, flat=None):
        flat = self.params if flat is None else flat
        out = {}
        for key, off, shape in self._spans:
            n = 1
            for s in shape:
                n *= s
            out[key] = flat[off:off + n].view(shape)
        return out

    def load(self, head):
        v = self.named_views()
        with torch.no_grad():
            for k, t in head.items():
                v[k].copy_(t.to(self.device, torch.float32).reshape(v[k].shape))
        self._stale = True

    def mark_stale(self):
        self._stale = True

    def is_stale(self):
        """True when the bf16 GEMM operands no longer match the fp32 masters: flagged explicitly (mark_stale) or the
        flat parameter buffer was written in place through any view since the last refresh - torch bumps the shared
        version counter for that, which is how an external `optimizer.step()` on the exposed nn.Parameters
        (train_negplus.py:121-123) is noticed without a parameters_changed() call."""
        return self._stale or self.params._version != getattr(self, "_seen_version", -1)

    def _refresh(self):
        v = self.named_views()
        self.wfc_b, self.wfc_t = ops.cast_transpose_bf16(v["text_fc.weight"])            # [C, C] and its transpose
        self._stale = False
        self._seen_version = self.params._version

    def _ws(self, B):
        return ops.workspace(lib().spn_tg_ws_bytes(B, self.C), self.device, "tgcir")

    def forward(self, feats, tokens, tokens_b, ref_tokens):
        """feats [B, C], tokens [B, L, C] (+ bf16 copy) from TextTower.forward_tokens, ref_tokens [B, NT, C] fp32
        -> pooled fused feature [B, C] (un-normalised); state kept for backward()."""
        if self.is_stale():
            self._refresh()
        B, L, C = tokens.shape
        S, G, NT = self.S, self.G, self.NT
        v = self.named_views()
        dev = self.device
        z = ops.gemm_nt(tokens_b.view(B * L, C), self.wfc_b, bias=v["text_fc.bias"], out_dtype=torch.float32)
        attn = torch.empty(B, L, S, dtype=torch.float32, device=dev)
        mod = torch.empty(B, NT, C, dtype=torch.float32, device=dev)
        check(lib().spn_tg_tokenlearn_fwd(_p(z), _p(v["tokenlearn_text.weight"]), _p(v["tokenlearn_text.bias"]), _p(attn),
                                          _p(mod), B, L, C, S, G, _stream()), "tg_tokenlearn_fwd")
        ref_tokens = ref_tokens.contiguous()
        xf = torch.empty(B * NT, 2 * C, dtype=torch.float32, device=dev)
        check(lib().spn_tg_fuse_prep(_p(feats), _p(v["masks_text.weight"]), _p(ref_tokens), _p(mod), None, _p(xf), B, C, S, G,
                                     _stream()), "tg_fuse_prep")
        # s_remain_map[0] forward on the fp32-exact GEMM (f32-input MFMA): its sign decides the ReLU mask of the
        # backward, and a bf16 pre-activation flips ~1 % of it (5 % error in this layer's gradient); B*NT x 2C x C
        # is ~3 GFLOP at B = 256, noise next to the tower
        hpre = torch.empty(B * NT, C, dtype=torch.float32, device=dev)
        w1 = v["s_remain_map.0.weight"]
        check(lib().spn_gemm_f32(_p(xf), _p(w1), B * NT, C, 2 * C, 2 * C, 2 * C, 0, _p(v["s_remain_map.0.bias"]), 0, None, 0,
                                 _p(hpre), C, 1.0, _stream()), "gemm_f32")
        remain = torch.empty(B, NT, dtype=torch.float32, device=dev)
        pooled = torch.empty(B, C, dtype=torch.float32, device=dev)
        check(lib().spn_tg_gate_fwd(_p(hpre), _p(v["s_remain_map.2.weight"]), _p(v["s_remain_map.2.bias"]), _p(ref_tokens),
                                    _p(mod), _p(remain), _p(pooled), B, NT, C, _stream()), "tg_gate_fwd")
        self._st = dict(feats=feats, tokens_b=tokens_b, z=z, attn=attn, mod=mod, ref=ref_tokens, xf=xf, hpre=hpre,
                        remain=remain, B=B, L=L)
        return pooled, mod

    def backward(self, dpooled):
        """-> (dfeats [B, C], dtokens [B, L, C]); fills self.grads (overwrites)."""
        st = self._st
        B, L, C, S, G, NT = st["B"], st["L"], self.C, self.S, self.G, self.NT
        v, g = self.named_views(), self.named_views(self.grads)
        dev, ws = self.device, self._ws(st["B"])
        dmod = torch.empty(B, NT, C, dtype=torch.float32, device=dev)
        dh = torch.empty(B * NT, C, dtype=torch.float32, device=dev)
        dht = torch.empty(C, B * NT, dtype=torch.float32, device=dev)
        check(lib().spn_tg_gate_bwd(_p(dpooled.contiguous()), _p(st["ref"]), _p(st["mod"]), _p(st["remain"]), _p(st["hpre"]),
                                    _p(v["s_remain_map.2.weight"]), _p(dmod), _p(dh), _p(dht), _p(g["s_remain_map.2.weight"]),
                                    _p(g["s_remain_map.0.bias"]), _p(g["s_remain_map.2.bias"]), _p(ws), ws.numel(), B, NT, C,
                                    _stream()), "tg_gate_bwd")
        # s_remain_map[0] on the fp32-exact GEMM: dW1 = dh^T X, dX = dh W1 (both operands [K, N]-major: b_is_kn = 1)
        w1 = v["s_remain_map.0.weight"]
        check(lib().spn_gemm_f32(_p(dht), _p(st["xf"]), C, 2 * C, B * NT, B * NT, 2 * C, 1, None, 0, None, 0,
                                 _p(g["s_remain_map.0.weight"]), 2 * C, 1.0, _stream()), "gemm_f32 dW1")
        dx = torch.empty(B * NT, 2 * C, dtype=torch.float32, device=dev)
        check(lib().spn_gemm_f32(_p(dh), _p(w1), B * NT, 2 * C, C, C, 2 * C, 1, None, 0, None, 0, _p(dx), 2 * C, 1.0,
                                 _stream()), "gemm_f32 dX")
        dfeats = torch.empty(B, C, dtype=torch.float32, device=dev)
        check(lib().spn_tg_mod_bwd(_p(dx), _p(dmod), _p(st["feats"]), _p(v["masks_text.weight"]), _p(dfeats),
                                   _p(g["masks_text.weight"]), _p(ws), ws.numel(), B, C, S, G, _stream()), "tg_mod_bwd")
        dz = torch.empty(B * L, C, dtype=torch.bfloat16, device=dev)
        check(lib().spn_tg_tokenlearn_bwd(_p(st["z"]), _p(v["tokenlearn_text.weight"]), _p(st["attn"]), _p(dmod), _p(dz),
                                          _p(g["tokenlearn_text.weight"]), _p(g["tokenlearn_text.bias"]), _p(ws), ws.numel(),
                                          B, L, C, S, G, _stream()), "tg_tokenlearn_bwd")
        # text_fc: dW = dz^T tokens, db = column sums of dz, dtokens = dz W
        _, dbfc = ops.gemm_tn(dz, st["tokens_b"].view(B * L, C), out=g["text_fc.weight"], want_colsum=True)
        g["text_fc.bias"].copy_(dbfc)
        dtokens = ops.gemm_nt(dz, self.wfc_t, out_dtype=torch.float32)
        return dfeats, dtokens.view(B, L, C)


class _TgcirStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, ids, ref_tokens, labels):
        feats, tokens, tokens_b = model.text.forward_tokens(ids)
        pooled, _ = model.head.forward(feats, tokens, tokens_b, ref_tokens)
        q, qb, inv = ops.combine_l2norm_fwd(None, None, pooled)
        bank = model._target_bank_dev
        saved = ops.bank_logits_buffer(qb.shape[0], bank.shape[0], qb.device)
        stats = ops.bank_stats_fwd(qb, bank, labels, 1.0 / model.tau, save=saved)
        lse, row, mean = ops.bank_loss_finalize(stats, bank.shape[0])
        ctx.model, ctx.st = model, dict(q=q, qb=qb, inv=inv, lse=lse, labels=labels, B=ids.shape[0], saved=saved)
        return mean.reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        m, st = ctx.model, ctx.st
        bank = m._target_bank_dev
        dq = ops.bank_grad_q(st["qb"], bank, st["labels"], 1.0 / m.tau, st["lse"], 1.0 / st["B"],
                             M_total=bank.shape[0], saved=st["saved"])[:, :m.head.C].contiguous()
        # autograd's d(loss) scales the (linear) backward on the device: no host synchronisation before the first launch
        scale = grad_out.detach().to(device=dq.device, dtype=torch.float32).reshape(1)
        dpooled = ops.combine_l2norm_bwd(st["q"], st["inv"], dq, scale=scale)
        snap_h = gradsink.snapshot(m._params, m.head.grads, m.head.named_views)
        snap_t = gradsink.snapshot(m._params, m.text.grads, m.text.named_views, "clip.")
        dfeats, dtokens = m.head.backward(dpooled)
        flat = m.text.backward_tokens(dfeats, dtokens)
        gradsink.publish(m._params, flat, m.text.named_views, snap_t, "clip.")
        gradsink.publish(m._params, m.head.grads, m.head.named_views, snap_h)
        return torch.zeros((), device=grad_out.device), None, None, None, None


class CIRPlus(nn.Module):
    """tgcir/models.py CIRPlus for the second stage.  `clip_model_name`: a CLIP state dict (or a path to one saved
    with torch.save) - `clip.load(name)` downloads are not available offline."""

    def __init__(self, clip_model_name, tau=0.01, transform="targetpad", target_ratio=1.25, device=torch.device("cuda"),
                 plus=False, local_token_num=8, global_token_num=4, tokenizer=None):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("spn4cir_amd runs on an MI355X (device='cuda'); there is no CPU path")
        if local_token_num != S_LOCAL:
            raise ValueError("the TokenLearner kernels are built for 8 local tokens (the reference default)")
        from .models import CIRPlus as _ClipCIRPlus
        sd = _ClipCIRPlus._resolve_state_dict(clip_model_name)     # dict, JIT archive / state-dict file, or cached model name
        c = text_cfg_from_state_dict(sd)
        if c["embed_dim"] != c["width"]:
            raise ValueError("TG-CIR feeds ln_final tokens and projected features to the same 512-d head: width == embed_dim")
        self.tau, self.plus = tau, plus
        self.text = TextTower(c["width"], c["layers"], c["heads"], c["embed_dim"], c["vocab"], c["ctx"], self.device)
        self.text.load_clip_state_dict(sd)
        self.head = TgcirHead(c["width"], global_token_num, self.device)
        self.output_dim = c["embed_dim"]
        self.tokenizer = tokenizer
        self.backbone = nn.Module()                 # parameter container: names as in the reference's state_dict()
        self._params = {}
        for key, view in self.text.named_views().items():
            p = nn.Parameter(view, requires_grad=True)
            self._register(self.backbone, "clip." + key, p)
            self._params["clip." + key] = p
        hv = self.head.named_views()
        for key in ("text_fc.weight", "text_fc.bias", "masks_text.weight"):
            p = nn.Parameter(hv[key], requires_grad=True)
            self._register(self.backbone, key, p)
            self._params[key] = p
        # tokenlearn_text is stored stacked ([S, C] / [S]); the reference keeps S Conv1d(C, 1, 1) modules
        # (backbone.tokenlearn_text.tokenizers.{s}.conv.0.{weight [1, C, 1], bias [1]}): see state_dict_reference()
        for key in ("tokenlearn_text.weight", "tokenlearn_text.bias"):
            p = nn.Parameter(hv[key], requires_grad=True)
            self._register(self.backbone, key, p)
            self._params[key] = p
        for key in ("s_remain_map.0.weight", "s_remain_map.0.bias", "s_remain_map.2.weight", "s_remain_map.2.bias"):
            p = nn.Parameter(hv[key], requires_grad=True)
            self._register(self, key, p)
            self._params[key] = p
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.refer_bank = None
        self._target_bank = self._target_bank_dev = None
        # frozen image side: CLIP ViT + fc (Linear(768, 512)) + TokenLearner + masks (models.py:55-75)
        self.vision, self.img_head, self.input_dim = None, None, None
        if "visual.proj" in sd and "visual.conv1.weight" in sd:
            vc = vision_cfg_from_state_dict(sd)
            self.vision = VisionTower(vc["width"], vc["layers"], vc["heads"], vc["patch"], vc["res"], vc["embed_dim"],
                                      self.device)
            self.vision.load_clip_state_dict(sd)
            self.input_dim = vc["res"]
            for key, view in self.vision.named_views().items():
                self._register(self.backbone, "clip.visual." + key, nn.Parameter(view, requires_grad=False))

    @staticmethod
    def _register(root, dotted, param):
        node = root
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, nn.Module())
            node = getattr(node, p)
        node.register_parameter(parts[-1], param)

    def load_head(self, head):
        """head: dict with the keys of oracle.tgcir_head.HEAD_KEYS (stacked TokenLearner weights)."""
        self.head.load(head)

    def load_img_head(self, ihead):
        """Frozen image-side head: dict with fc.weight [C, Wv], fc.bias, tokenlearn.weight [S, C] (stacked Conv1d
        kernels), tokenlearn.bias [S], masks.weight [G, C] (oracle.tgcir_head.IMG_HEAD_KEYS)."""
        h = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in ihead.items()}
        h["fc.weight_bf16"] = ops.cast_bf16(h["fc.weight"])
        self.img_head = h

    def load_reference_state_dict(self, sd):
        """Accepts tgcir's own `state_dict()` naming (checkpoint['state_dict'], models.py:207-209)."""
        C, S = self.head.C, self.head.S
        head = {"text_fc.weight": sd["backbone.text_fc.weight"], "text_fc.bias": sd["backbone.text_fc.bias"],
                "masks_text.weight": sd["backbone.masks_text.weight"],
                "tokenlearn_text.weight": torch.stack([sd[f"backbone.tokenlearn_text.tokenizers.{s}.conv.0.weight"].reshape(C)
                                                       for s in range(S)]),
                "tokenlearn_text.bias": torch.cat([sd[f"backbone.tokenlearn_text.tokenizers.{s}.conv.0.bias"].reshape(1)
                                                   for s in range(S)])}
        for k in ("s_remain_map.0.weight", "s_remain_map.0.bias", "s_remain_map.2.weight", "s_remain_map.2.bias"):
            head[k] = sd[k]
        self.head.load(head)
        self.text.load_clip_state_dict(sd, prefix="backbone.clip.")
        if "backbone.fc.weight" in sd:
            self.load_img_head({"fc.weight": sd["backbone.fc.weight"], "fc.bias": sd["backbone.fc.bias"],
                                "masks.weight": sd["backbone.masks.weight"],
                                "tokenlearn.weight": torch.stack([sd[f"backbone.tokenlearn.tokenizers.{s}.conv.0.weight"].reshape(C)
                                                                  for s in range(S)]),
                                "tokenlearn.bias": torch.cat([sd[f"backbone.tokenlearn.tokenizers.{s}.conv.0.bias"].reshape(1)
                                                              for s in range(S)])})
            if self.vision is not None and "backbone.clip.visual.proj" in sd:
                self.vision.load_clip_state_dict(sd, prefix="backbone.clip.visual.")

    def load_ckpt(self, model_path, is_origin=False):
        saved = torch.load(model_path, map_location="cpu")
        self.load_reference_state_dict(saved["state_dict"])
        if is_origin:
            # models.py:210-213: the text-side TokenLearner / masks start as copies of the image-side ones
            sd, C, S = saved["state_dict"], self.head.C, self.head.S
            self.head.load({"masks_text.weight": sd["backbone.masks.weight"],
                            "tokenlearn_text.weight": torch.stack([sd[f"backbone.tokenlearn.tokenizers.{s}.conv.0.weight"].reshape(C)
                                                                   for s in range(S)]),
                            "tokenlearn_text.bias": torch.cat([sd[f"backbone.tokenlearn.tokenizers.{s}.conv.0.bias"].reshape(1)
                                                               for s in range(S)])})

    def parameters_changed(self):
        """Call after an optimizer step that wrote the parameters (refreshes the bf16 GEMM operands lazily)."""
        self.text.mark_stale()
        self.head.mark_stale()

    # ------------------------------------------------------------------------------- banks
    @property
    def target_bank(self):
        return self._target_bank

    @target_bank.setter
    def target_bank(self, bank):
        self._target_bank = bank
        self._target_bank_dev = None if bank is None else ops.prepare_bank(bank.to(self.device, torch.float32))

    def img_embed(self, image, return_pool_and_normalized=False):
        """models.py:183-196: image tokens [B, 12, C] (4 masked global + 8 TokenLearner tokens); with the flag also the
        normalised mean over the tokens (the target-bank row)."""
        if self.vision is None or self.img_head is None:
            raise RuntimeError("image side needs a ViT state dict (visual.*) and load_img_head() / a reference checkpoint")
        h, C, S, G = self.img_head, self.head.C, self.head.S, self.head.G
        with torch.no_grad():
            feats, vt = self.vision.forward(image.to(self.device, torch.float32), return_tokens=True)
            B, L, Wv = vt.shape
            z = ops.gemm_nt(ops.cast_bf16(vt.view(B * L, Wv)), h["fc.weight_bf16"], bias=h["fc.bias"], out_dtype=torch.float32)
            attn = torch.empty(B, L, S, dtype=torch.float32, device=self.device)
            tokens = torch.empty(B, G + S, C, dtype=torch.float32, device=self.device)
            pooled = torch.empty(B, C, dtype=torch.float32, device=self.device)
            check(lib().spn_tg_tokenlearn_fwd(_p(z), _p(h["tokenlearn.weight"]), _p(h["tokenlearn.bias"]), _p(attn), _p(tokens),
                                              B, L, C, S, G, _stream()), "tg_tokenlearn_fwd")
            check(lib().spn_tg_img_finish(_p(feats), _p(h["masks.weight"]), _p(tokens), _p(pooled), B, C, S, G, _stream()),
                  "tg_img_finish")
            if return_pool_and_normalized:
                return [tokens, ops.combine_l2norm_fwd(None, None, pooled)[0]]
        return tokens

    @staticmethod
    def _image_batches(dataset, bs=128, decode_bs=1024):
        """Items of the dataset in batches of `bs` for the image tower.  The items are fetched `decode_bs` at a time with the
        dataset's transform in deferred mode: undecoded JPEGs come back as file bytes and the whole chunk (reference AND target
        images) is decoded on the GPU in one call - a lane per file, so the chunk size is the decoder's parallelism."""
        n = len(dataset)
        for s in range(0, n, decode_bs):
            with gpu_decode_scope(dataset):
                items = [dataset[i] for i in range(s, min(n, s + decode_bs))]
            items = realize_items([it for it in items if it is not None])      # utils.collate_fn drops None samples
            for k in range(0, len(items), bs):
                yield items[k:k + bs]

    def extract_bank_features(self, cirDataset, device=None, bank_path=None, reload_bank=False):
        """models.py:223-250: per-triplet reference token bank [len, 12, 512] + normalised pooled target bank."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            self.refer_bank, self.target_bank = torch.load(bank_path)
            return
        NT, C = self.head.NT, self.head.C
        refer, target = torch.zeros(len(cirDataset), NT, C), torch.zeros(cirDataset.image_id, C)
        for items in self._image_batches(cirDataset):
            rtok, rpool = self.img_embed(stack_images([it[0] for it in items]), True)
            _, tpool = self.img_embed(stack_images([it[2] for it in items]), True)
            refer[torch.tensor([int(it[3]) for it in items])] = rtok.cpu()
            target[torch.tensor([int(it[5]) for it in items])] = rpool.cpu()
            target[torch.tensor([int(it[6]) for it in items])] = tpool.cpu()
        self.refer_bank, self.target_bank = refer, target
        if bank_path:
            torch.save([refer, target], bank_path)

    def extract_refer_bank_features(self, cirDataset, device=None, bank_path=None, reload_bank=False):
        """models.py:252-267 (--plus): token bank per unique image id, written to bank_path (read back by
        load_refer_bank, as in the reference)."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            return
        refer = torch.zeros(cirDataset.image_id, self.head.NT, self.head.C)
        for items in self._image_batches(cirDataset):
            refer[torch.tensor([int(it[5]) for it in items])] = self.img_embed(stack_images([it[0] for it in items])).cpu()
            refer[torch.tensor([int(it[6]) for it in items])] = self.img_embed(stack_images([it[2] for it in items])).cpu()
        self.refer_bank = refer
        if bank_path:
            torch.save(refer, bank_path)

    def load_refer_bank(self, bank_path):
        self.refer_bank = torch.load(bank_path)

    # -------------------------------------------------------------------------------- step
    def tokenize(self, text):
        if torch.is_tensor(text):
            ids = text
        else:
            if self.tokenizer is None:
                from .tokenizer import tokenize
                self.tokenizer = tokenize
            ids = self.tokenizer(list(text))
        return ids.to(self.device, torch.int32).contiguous()

    def extract_text_fea(self, text):
        """Backbone.extract_text_fea (models.py:127-151) -> mod tokens [B, 12, C] (inference form)."""
        ids = self.tokenize(text)
        with torch.no_grad():
            feats, tokens, tokens_b = self.text.forward_tokens(ids)
            dummy = torch.zeros(ids.shape[0], self.head.NT, self.head.C, device=self.device)
            _, mod = self.head.forward(feats, tokens, tokens_b, dummy)
        return mod

    def img_txt_fusion(self, ref_token, mod):
        """models.py:198-205 (inference form): normalised pooled fusion of reference tokens and the modifier text."""
        ids = self.tokenize(mod)
        with torch.no_grad():
            feats, tokens, tokens_b = self.text.forward_tokens(ids)
            pooled, _ = self.head.forward(feats, tokens, tokens_b, ref_token.to(self.device, torch.float32))
        return torch.nn.functional.normalize(pooled, dim=-1)

    def forward(self, text, indexs, target_indexs, refer_indexs, refer_image=None, target_image=None):
        """models.py:272-289 -> {'bank_loss': 0-dim tensor}; backward() fills .grad of every second-stage parameter."""
        ids = self.tokenize(text)
        idx = refer_indexs if self.plus else indexs
        ref = self.refer_bank[idx.to(self.refer_bank.device)].to(self.device, torch.float32)
        labels = target_indexs.to(self.device, torch.int64)
        return {"bank_loss": _TgcirStep.apply(self._anchor, self, ids, ref, labels)}


class TgcirStage2Trainer:
    """tgcir/train.py's second-stage loop body (:82-92 optimizer, :120-135 step) on one GPU or data-parallel over
    RCCL: triplets sharded across ranks, text tower + head replicated, bank replicated or sharded as for the CLIP path
    (spn4cir_amd.distributed.BankLossDP), both flat gradient buffers all-reduced, one fused AdamW launch per buffer
    (lr, betas (0.9, 0.999), eps 1e-7, torch's default weight decay - what optim.AdamW(param_groups) configures)."""

    def __init__(self, model, lr=5e-6, betas=(0.9, 0.999), eps=1e-7, weight_decay=0.01, group=None, bank_mode="auto",
                 label_smoothing=0.0, bucket_elems=8 << 20):
        from . import distributed as dp
        self.model, self.group = model, group
        self._bank_mode_arg = bank_mode      # "auto": replicated below 10^6 bank rows (set_banks), as bench.py --bank-mode auto
        bank_mode = "replicated" if bank_mode == "auto" else bank_mode
        self.text, self.head = model.text, model.head
        self.lr, self.betas, self.eps, self.wd, self.ls = lr, betas, eps, weight_decay, label_smoothing
        self.world, self.rank = dp._world(group)
        self.loss_dp = dp.BankLossDP(ops, group, bank_mode if (self.world > 1 or dp._FORCE) else "replicated")
        self.red_text = dp.GradBucketReducer(self.text.grads, group, bucket_elems)
        self.red_head = dp.GradBucketReducer(self.head.grads, group, bucket_elems)
        self._shard_range = dp.shard_range
        self.state = [(torch.zeros_like(self.text.params), torch.zeros_like(self.text.params)),
                      (torch.zeros_like(self.head.params), torch.zeros_like(self.head.params))]
        self.step_count = 0
        self._bank, self._m_begin, self._M_total = None, 0, 0
        self.refer_bank = None

    def set_banks(self, refer_bank, target_bank):
        """refer_bank [N, 12, C] token bank (kept on the device), target_bank fp32 [M, C] normalised rows; the sharded
        mode keeps only this rank's target rows."""
        dev = self.text.device
        self.refer_bank = refer_bank.to(dev, torch.float32).contiguous()
        self._M_total = target_bank.shape[0]
        if self._bank_mode_arg == "auto" and self.world > 1:
            self.loss_dp.mode = "sharded" if self._M_total >= 1000000 else "replicated"
        if self.loss_dp.mode == "sharded" and self.world > 1:
            b, e = self._shard_range(self._M_total, self.world, self.rank)
            self._m_begin = b
            self._bank = ops.prepare_bank(target_bank[b:e].to(dev, torch.float32).contiguous())
        else:
            self._m_begin = 0
            self._bank = ops.prepare_bank(target_bank.to(dev, torch.float32))

    def step(self, ids, refer_idx, labels):
        """ids int32 [B_local, L], refer_idx int64 [B_local] rows of the token bank, labels int64 [B_local] global
        target rows (all on the device).  Returns the global mean loss (1-element device tensor)."""
        text, head = self.text, self.head
        feats, tokens, tokens_b = text.forward_tokens(ids)
        pooled, _ = head.forward(feats, tokens, tokens_b, self.refer_bank[refer_idx])
        q, qb, inv = ops.combine_l2norm_fwd(None, None, pooled)
        ctx = self.loss_dp.forward(qb, labels, self._bank, self._m_begin, self._M_total, 1.0 / self.model.tau, self.ls)
        dq = self.loss_dp.backward(ctx)[:, :head.C].contiguous()
        dfeats, dtokens = head.backward(ops.combine_l2norm_bwd(q, inv, dq))
        self.red_head.on_span_ready(0, head.n_params)
        if self.world > 1:
            # layer groups (7 + 3 + 2 of 12): a group's gradient range is all-reduced while the blocks below still run
            from .trainer import wgrad_groups
            text.backward_tokens_phased(dfeats, dtokens, self.red_text.on_span_ready, wgrad_groups(text.layers, self.world))
        else:
            text.backward_tokens(dfeats, dtokens)
            self.red_text.on_span_ready(0, text.n_params)
        self.red_head.finish()
        self.red_text.finish()
        self.step_count += 1
        for (p, g), (m, v) in zip(((text.params, text.grads), (head.params, head.grads)), self.state):
            ops.adamw_step(p, g, m, v, self.step_count, self.lr, self.betas, self.eps, self.wd)
        text.mark_stale()
        head.mark_stale()
        return ctx["loss"]
