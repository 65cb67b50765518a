"""blip4cir `CIRPlus` protocol (blip4cir/models.py:16-121) over the MI355X path: fusion encoder (med.py BertModel
+ text_proj), BLIP ViT image side, static token / target banks, learnable temperature.

    model = CIRPlus(blip_state_dict_or_path, tau=0.03, vocab_file="bert-base-uncased/vocab.txt")
    loss = model.forward(text, indexs, target_indexs, refer_indexs)['bank_loss']; loss.backward()
    extract_index_features(classic_val_dataset, model.blip); compute_fiq_val_metrics(rel_ds, model.blip, ...)

`text` is a list of strings: tokenised as blip_cir.py:87-88 does (WordPiece, padding='longest', first id <- [ENC]) by
`bert_tokenizer.BertWordPieceTokenizer` built from `vocab_file` (or $SPN_BERT_VOCAB; the bert-base-uncased vocab.txt
is a download this repository cannot vendor), or by `tokenizer=` - an object with the transformers call shape
(`tok(texts, padding='longest', return_tensors='pt')` + `.enc_token_id`, e.g. the reference's own `init_tokenizer()`)
or a plain callable `texts -> (ids [B,L], attention_mask [B,L])`.  An already tokenised `(ids, mask)` pair is taken as
is.  `model.blip` is the object the reference's loop hands to `extract_index_features` / `compute_*_val_metrics`
(blip4cir/train.py:59,70,134,167): it answers `img_embed`, `img_embed_p`, `img_txt_fusion`, `tokenizer`, `eval()` /
`train()` and holds the parameters under BLIP_Retrieval's names: `visual_encoder.*`, `vision_proj.*`,
`text_encoder.*`, `text_proj.*`, `temp` (under `blip.` in CIRPlus.state_dict())."""
import os
import weakref

import torch
from torch import nn

from . import gradsink, ops
from .preprocess import gpu_decode_scope, realize_items, stack_images
from .fusion import FusionEncoder, fusion_cfg_from_state_dict
from .vision_tower import VisionTower


class _FusionBankStep(torch.autograd.Function):
    """blip4cir/models.py:95-121 as one autograd node.  The learnable temperature (models.py:29) is READ ON THE DEVICE, as in
    fusion.BlipStage2Trainer: logits = (q / tau) . bank with the bank kernels at inv_tau = 1 and every factor of tau applied by
    device-side scalars.  (Round 5 read `float(tau)` here: a host synchronisation in front of every forward, behind the
    optimizer's update of tau - the device drained and then idled while the host tokenised and enqueued the step.)"""

    @staticmethod
    def forward(ctx, anchor, tau_param, model, ids, mask, token_bank, token_idx, labels):
        enc = model.fusion
        tau_dev = tau_param.detach().reshape(1)
        proj = enc.forward(ids, mask, token_bank=token_bank, token_idx=token_idx)
        q, _, inv = ops.combine_l2norm_fwd(None, None, proj)
        bank = model._target_bank_dev
        qs = ops.scale_cast_bf16(q, tau_dev, reciprocal=True, ldo=bank.shape[1])        # bf16(q / tau)
        saved = ops.bank_logits_buffer(qs.shape[0], bank.shape[0], qs.device)
        stats = ops.bank_stats_fwd(qs, bank, labels, 1.0, save=saved)
        lse, row, mean = ops.bank_loss_finalize(stats, bank.shape[0])
        ctx.model, ctx.st = model, dict(q=q, qs=qs, inv=inv, lse=lse, labels=labels, B=ids.shape[0], saved=saved, tau_dev=tau_dev)
        return mean.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        m, st = ctx.model, ctx.st
        enc = m.fusion
        bank = m._target_bank_dev
        # the incoming d(loss) scales the (linear) backward on the device: no host synchronisation on it
        scale = grad_out.detach().to(torch.float32).reshape(1)
        dqk = ops.bank_grad_q(st["qs"], bank, st["labels"], 1.0, st["lse"], 1.0 / st["B"], M_total=bank.shape[0],
                              saved=st["saved"])                                          # d loss / d (q / tau)
        # models.py:29: tau is an nn.Parameter.  d loss / d tau = -(sum q . dqk) / tau^2 times autograd's incoming scalar, and
        # 1 / tau for the chain into q - one launch, tau read on the device
        dtau = torch.empty(1, dtype=torch.float32, device=dqk.device)
        inv_tau = ops.tau_grad(st["q"], dqk, st["tau_dev"], dtau, scale_dev=scale)
        snap = gradsink.snapshot(m._params, enc.grads, enc.named_views)
        dqc = dqk if dqk.shape[1] == enc.Dp else dqk[:, :enc.Dp].contiguous()
        flat = enc.backward(ops.combine_l2norm_bwd(st["q"], st["inv"], dqc, scale=inv_tau * scale))
        gradsink.publish(m._params, flat, enc.named_views, snap)
        return None, dtau.reshape(()), None, None, None, None, None, None


class BlipRetrievalFacade(nn.Module):
    """`model.blip`: BLIP_Retrieval's inference surface (blip_cir.py:54-103) over the owning CIRPlus' kernels, and the
    parameter container whose names equal BLIP_Retrieval.state_dict()'s.  The owner is held weakly (it registers this
    module as a child; a strong reference back would make the module tree cyclic)."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner_ref", weakref.ref(owner))
        self.temp = nn.Parameter(0.07 * torch.ones([], device=owner.device), requires_grad=False)   # blip_cir.py:47; stage 1 only

    def _owner(self):
        owner = self._owner_ref()
        if owner is None:
            raise RuntimeError("the CIRPlus that owns this BLIP module is gone")
        return owner

    @property
    def tokenizer(self):
        return self._owner().tokenizer

    def init_stage2(self):                           # blip_cir.py:48-52: the image side is frozen here from the start
        for name, p in self.named_parameters():
            if name.startswith(("visual_encoder.", "vision_proj.")):
                p.requires_grad = False

    def img_embed(self, image, atts=False, return_pool_and_normalized=False):
        """blip_cir.py:54-70: tokens [B, S, W] (+ normalised vision_proj of token 0) (+ all-ones attention mask)."""
        owner = self._owner()
        tokens, pooled = owner._img_embed(image)
        out = (tokens,)
        if return_pool_and_normalized:
            out += (pooled,)
        if atts:
            out += (torch.ones(tokens.shape[:-1], dtype=torch.long, device=tokens.device),)
        return out[0] if len(out) == 1 else out

    def img_embed_p(self, image):
        """blip_cir.py:72-80."""
        return self._owner()._img_embed(image)[1]

    def img_txt_fusion(self, r_image_embeds, t_image_embeds, text, train=False, return_raw=False):
        """blip_cir.py:82-103, inference form.  `train=True` (B x B logits against in-batch targets over `temp`) is the
        first-stage loss of the candidate-reranking code this file was taken from; the second stage never calls it."""
        if train or return_raw:
            raise NotImplementedError("img_txt_fusion(train=True / return_raw=True) is not on the second-stage path "
                                      "(blip4cir/models.py:101 calls it with the defaults)")
        return self._owner()._fuse(r_image_embeds, text)


class CIRPlus(nn.Module):
    def __init__(self, blip_model_name, tau=0.01, transform="targetpad", target_ratio=1.25, encoder="both",
                 device=torch.device("cuda"), plus=False, tokenizer=None, image_size=384, patch=16, enc_token_id=None,
                 vocab_file=None):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("spn4cir_amd runs on an MI355X (device='cuda'); there is no CPU path")
        sd = blip_model_name
        if isinstance(sd, str):
            if not os.path.isfile(sd):
                raise RuntimeError(f"BLIP checkpoint {sd} not found")
            sd = torch.load(sd, map_location="cpu")
            sd = sd.get("model", sd.get("BLIP_Retrieval", sd))
        vocab_file = vocab_file or os.environ.get("SPN_BERT_VOCAB")
        if tokenizer is None and vocab_file:
            from .bert_tokenizer import init_tokenizer
            tokenizer = init_tokenizer(vocab_file)                       # blip.py:189-194 from a local vocab.txt
        if enc_token_id is None:
            enc_token_id = getattr(tokenizer, "enc_token_id", None)
        self.plus, self.encoder, self.tokenizer, self.enc_token_id = plus, encoder, tokenizer, enc_token_id
        c = fusion_cfg_from_state_dict(sd, "text_encoder.")
        self.output_dim = sd["text_proj.weight"].shape[0]
        self.fusion = FusionEncoder(c["hidden"], c["layers"], c["heads"], c["intermediate"], c["enc_width"], self.output_dim,
                                    c["vocab"], c["max_pos"], self.device)
        fsd = {k[len("text_encoder."):]: v for k, v in sd.items() if k.startswith("text_encoder.")}
        fsd.update({k: v for k, v in sd.items() if k.startswith("text_proj.")})
        self.fusion.load_state_dict(fsd)
        self.vision = None
        if "visual_encoder.cls_token" in sd:
            W = sd["visual_encoder.cls_token"].shape[-1]
            layers = len({k.split(".")[2] for k in sd if k.startswith("visual_encoder.blocks.")})
            patch = sd["visual_encoder.patch_embed.proj.weight"].shape[-1]
            grid = round((sd["visual_encoder.pos_embed"].shape[1] - 1) ** 0.5)
            image_size = patch * grid
            self.vision = VisionTower(W, layers, W // 64, patch, image_size, self.output_dim, self.device, kind=1)
            self.vision.load_blip_state_dict(sd)
        self.input_dim = image_size
        self.tau = nn.Parameter(tau * torch.ones([], device=self.device))
        self.blip = BlipRetrievalFacade(self)        # names as in BLIP_Retrieval.state_dict() + its inference methods
        self._params = {}
        for key, view in self.fusion.named_views().items():
            name = key if key.startswith("text_proj.") else "text_encoder." + key
            p = nn.Parameter(view, requires_grad=True)
            self._register(name, p)
            self._params[key] = p
        if self.vision is not None:
            for key, view in self.vision.named_views().items():
                if key == "vision_proj_t":
                    continue                          # exposed through the towers; stored transposed internally
                name = key if key.startswith("vision_proj.") else "visual_encoder." + key
                self._register(name, nn.Parameter(view, requires_grad=False))
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.preprocess = None
        if transform == "targetpad":
            from .preprocess import TargetPadTransform
            self.preprocess = TargetPadTransform(target_ratio, self.input_dim, self.device)
        self._refer_bank = self._refer_bank_dev = None
        self._target_bank = self._target_bank_dev = None

    def _register(self, dotted, param):
        node = self.blip
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, nn.Module())
            node = getattr(node, p)
        node.register_parameter(parts[-1], param)

    # ------------------------------------------------------------------------------- banks
    @property
    def target_bank(self):
        return self._target_bank

    @target_bank.setter
    def target_bank(self, bank):
        self._target_bank = bank
        self._target_bank_dev = None if bank is None else ops.prepare_bank(bank.to(self.device, torch.float32))

    @property
    def refer_bank(self):
        return self._refer_bank

    @refer_bank.setter
    def refer_bank(self, bank):
        """[N, S, W] reference-token bank (fp32 in host RAM in the reference, blip4cir/models.py:47,76).  The training step reads
        it from a bf16 image on the device, built on the first forward() after an assignment (26.6 GB at 30 000 x 577 x 768;
        ops.token_bank_bf16 uploads in chunks) - the reference gathers on the host and uploads 227 MB per step (models.py:97-100)."""
        self._refer_bank = bank
        self._refer_bank_dev = None

    def load_refer_bank(self, bank_path):
        self.refer_bank = torch.load(bank_path)

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

    def _embed(self, images):
        """img_embed + normalised vision_proj of token 0 for a list of [3, H, H] tensors."""
        tokens, pooled = self.img_embed(stack_images(images).to(self.device, torch.float32), return_pool_and_normalized=True)
        return tokens, pooled

    def extract_bank_features(self, cirDataset, device=None, bank_path=None, reload_bank=False):
        """blip4cir/models.py:45-71: per-triplet reference TOKEN bank [len, S, W] (577 x 768 for ViT-B/16@384) and
        the normalised pooled target bank [image_id, output_dim]; same `torch.save([refer_bank, target_bank])` file."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            self.refer_bank, self.target_bank = torch.load(bank_path)
            return
        refer, target = None, torch.zeros(cirDataset.image_id, self.output_dim)
        for items in self._image_batches(cirDataset):
            rtok, rpool = self._embed([it[0] for it in items])
            _, tpool = self._embed([it[2] for it in items])
            if refer is None:
                refer = torch.zeros(len(cirDataset), rtok.shape[1], rtok.shape[2])
            refer[torch.tensor([int(it[3]) for it in items])] = rtok.cpu()
            target[torch.tensor([int(it[5]) for it in items])] = rpool.cpu()
            target[torch.tensor([int(it[6]) for it in items])] = tpool.cpu()
        self.refer_bank, self.target_bank = refer, target
        if bank_path:
            torch.save([refer, target], bank_path)

    def extract_refer_bank_features(self, cirDataset, device=None, bank_path=None, reload_bank=False):
        """blip4cir/models.py:73-89 (--plus): token bank per unique image id; written to bank_path, read back with
        load_refer_bank (the reference never loads it here either)."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            return
        refer = None
        for items in self._image_batches(cirDataset):
            rtok, _ = self._embed([it[0] for it in items])
            ttok, _ = self._embed([it[2] for it in items])
            if refer is None:
                refer = torch.zeros(cirDataset.image_id, rtok.shape[1], rtok.shape[2])
            refer[torch.tensor([int(it[5]) for it in items])] = rtok.cpu()
            refer[torch.tensor([int(it[6]) for it in items])] = ttok.cpu()
        self.refer_bank = refer
        if bank_path:
            torch.save(refer, bank_path)

    def _img_embed(self, image):
        if self.vision is None:
            raise RuntimeError("this checkpoint has no visual_encoder")
        with torch.no_grad():
            feats, tokens = self.vision.forward(image.to(self.device, torch.float32), return_tokens=True)
            return tokens, ops.combine_l2norm_fwd(None, None, feats.contiguous())[0]

    def img_embed(self, image, return_pool_and_normalized=False):
        """BLIP_Retrieval.img_embed (blip_cir.py:54-70): token sequence [B, S, W] (and normalised vision_proj of token 0)."""
        tokens, pooled = self._img_embed(image)
        return (tokens, pooled) if return_pool_and_normalized else tokens

    # -------------------------------------------------------------------------------- step
    def tokenize(self, text):
        if isinstance(text, (tuple, list)) and len(text) == 2 and torch.is_tensor(text[0]):
            ids, mask = text
        else:
            if self.tokenizer is None:
                raise RuntimeError("no tokenizer: construct CIRPlus with vocab_file=<bert-base-uncased vocab.txt> (or set "
                                   "SPN_BERT_VOCAB), pass tokenizer=, or feed (ids, mask)  (blip_cir.py:87-88)")
            texts = [text] if isinstance(text, str) else list(text)
            out = self.tokenizer(texts, padding="longest", return_tensors="pt") if hasattr(self.tokenizer, "enc_token_id") \
                else self.tokenizer(texts)
            ids, mask = (out["input_ids"], out["attention_mask"]) if hasattr(out, "keys") else out
        ids = ids.to(self.device, torch.int32).clone()
        if self.enc_token_id is not None:
            ids[:, 0] = self.enc_token_id              # blip_cir.py:88
        # the mask stays where it came from: a host mask (the tokenizer's) lets the encoder drop the padded rows (FusionEncoder.forward)
        return ids.contiguous(), mask.to(torch.int32).contiguous()

    def _fuse(self, r_image_embeds, text):
        ids, mask = self.tokenize(text)
        with torch.no_grad():
            proj = self.fusion.forward(ids, mask, r_image_embeds.to(self.device, torch.float32))
            return ops.combine_l2norm_fwd(None, None, proj)[0]

    def img_txt_fusion(self, r_image_embeds, t_image_embeds, text):
        """Inference form (blip_cir.py:82-103, train=False): normalised text_proj of the fused [ENC] token."""
        return self._fuse(r_image_embeds, text)

    def forward(self, text, indexs, target_indexs, refer_indexs, reference_image=None, target_image=None):
        """blip4cir/models.py:95-110 -> {'bank_loss': 0-dim tensor}; backward() fills .grad of the fusion encoder's
        parameters and of `tau`."""
        ids, mask = self.tokenize(text)
        idx = refer_indexs if self.plus else indexs
        if self._refer_bank is None:
            raise RuntimeError("no reference-token bank: extract_bank_features / load_refer_bank first (train.py:103-108)")
        ops.check_index_range(idx, self._refer_bank.shape[0], "refer_indexs" if self.plus else "indexs")
        if self._refer_bank_dev is None:
            self._refer_bank_dev = ops.token_bank_bf16(self._refer_bank, self.device)
        labels = target_indexs.to(self.device, torch.int64)
        loss = _FusionBankStep.apply(self._anchor, self.tau, self, ids, mask, self._refer_bank_dev,
                                     idx.to(self.device, torch.int64), labels)
        return {"bank_loss": loss}

    def parameters_changed(self):
        self.fusion.mark_stale()

    def load_ckpt(self, model_path, is_origin=False):
        saved = torch.load(model_path, map_location="cpu")
        src = {"blip." + k: v for k, v in saved["BLIP_Retrieval"].items()} if is_origin else saved["state_dict"]
        own = self.state_dict()
        with torch.no_grad():
            for k, v in src.items():
                if k in own and own[k].shape == v.shape:
                    own[k].copy_(v.to(own[k].device, own[k].dtype))
        self.fusion.mark_stale()
        if self.vision is not None:
            self.vision.mark_stale()
