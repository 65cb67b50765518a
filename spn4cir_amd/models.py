"""`CIRPlus`: the reference's stage-2 model protocol on the MI355X kernels.

Mirrors clip4cir/models_negplus.py:16-154 (and the zero-shot variant zscir/models_bank.py:18-134):
same constructor arguments, attributes (`tau`, `device`, `output_dim`, `input_dim`, `refer_bank`,
`target_bank`, `M`, `clip`), methods (`encode_text`, `combining_function`, `forward ->
{'bank_loss'}`, `load_ckpt`, `load_refer_bank`, bank setters) and state-dict keys (`clip.*`).

Differences that do not change results: the target bank lives on the
device as bf16 for the whole run instead of being copied host->device every step, and the
B x M logits are never materialised.

The device work is the C-ABI of include/spn4cir_hip.h; there is no CPU fallback.
"""
import os

import torch
from torch import nn

from . import gradsink, ops
from .preprocess import gpu_decode_scope, realize_items, stack_images
from .text_tower import TextTower, text_cfg_from_state_dict
from .vision_tower import VisionTower, vision_cfg_from_state_dict


class _BankStep(torch.autograd.Function):
    """loss = mean CE((normalize(ref + text_tower(ids)) @ bank.T) / tau, labels).

    Forward enqueues tower fwd + combiner + bank statistics; backward enqueues the bank gradient,
    combiner and tower backward and leaves the flat gradient in `model.tower.grads`
    (parameters' `.grad` are views of it).  `anchor` only exists so autograd calls backward()."""

    @staticmethod
    def forward(ctx, anchor, model, ids, refer_idx, labels):
        st = model._step_forward(ids, refer_idx, labels)
        ctx.model, ctx.st = model, st
        return st["loss"].reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.model._step_backward(ctx.st, grad_out)
        return torch.zeros((), device=grad_out.device), None, None, None, None


class _InBatchStep(torch.autograd.Function):
    """BASELINE config 1: loss = CE(normalize(vis(ref) + text(ids)) @ normalize(vis(tgt)).T / tau, arange(B))."""

    @staticmethod
    def forward(ctx, anchor, model, ids, refer_image, target_image):
        st = model._inbatch_forward(ids, refer_image, target_image)
        ctx.model, ctx.st = model, st
        return st["loss"].reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.model._inbatch_backward(ctx.st, grad_out)
        return torch.zeros((), device=grad_out.device), None, None, None, None


class _Tree(nn.Module):
    """Nested containers so that parameter names reproduce the reference's dotted keys."""


def _attach(root, dotted, param):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Tree())
        node = getattr(node, p)
    node.register_parameter(parts[-1], param)


# clip/clip.py:30-40: model name -> file name of the published archive
_CLIP_FILES = {"RN50": "RN50.pt", "RN101": "RN101.pt", "RN50x4": "RN50x4.pt", "RN50x16": "RN50x16.pt", "RN50x64": "RN50x64.pt",
               "ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt",
               "ViT-L/14@336px": "ViT-L-14-336px.pt"}


class CIRPlus(nn.Module):
    def __init__(self, clip_model_name, tau=0.01, transform="targetpad", target_ratio=1.25,
                 device=torch.device("cuda"), plus=False, neg_num=-1, combiner="sum", label_smoothing=0.0,
                 tokenizer=None, pack_eot=True, wo_bank=False, exact_eval=True):
        """`clip_model_name`: path to a CLIP state-dict file (as clip.load accepts, clip/clip.py:120-123),
        a state-dict, or "synthetic:<name>" (seeded random weights; no pretrained weights exist offline).
        `pack_eot` (default since round 5): run the text tower on the live rows only (everything after a caption's EOT token is
        dead under the causal mask, clip/model.py:330-336,356): bit-identical features, the same loss and gradients, ~L / mean_len
        fewer rows (2.4x the step rate on 5-30 word captions).  It needs the ids on the HOST (strings, or a CPU id tensor - what
        the reference's DataLoader hands over): the live lengths then cost no device sync; ids that already sit on the device run
        dense.  pack_eot=False always runs all 77 positions."""
        super().__init__()
        self.pack_eot = bool(pack_eot)
        # exact_eval (default): encode_image / encode_text under torch.no_grad() - validation, bank extraction - run the
        # fp32-exact towers, whose features give the reference's top-K index sets (north_star); False = the bf16 towers
        # there too (8x the throughput, ~1e-5 feature error: enough to flip near-ties of a ranking).  Training is bf16.
        self.exact_eval = bool(exact_eval)
        self.wo_bank = bool(wo_bank)   # clip4cir/models.py:23: in-batch negatives, visual tower trainable
        self._pack = (None, 0)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("spn4cir_amd.CIRPlus runs on an MI355X (device='cuda'); there is no CPU path")
        sd = self._resolve_state_dict(clip_model_name)
        cfg = text_cfg_from_state_dict(sd)
        self.tower = TextTower(cfg["width"], cfg["layers"], cfg["heads"], cfg["embed_dim"], cfg["vocab"], cfg["ctx"],
                               self.device)
        self.tower.load_clip_state_dict(sd)
        # parameters = views of the flat buffer, registered under the reference's names
        self.clip = _Tree()
        self._params = {}
        for key, view in self.tower.named_views().items():
            p = nn.Parameter(view, requires_grad=True)
            _attach(self.clip, key, p)
            self._params[key] = p
        self.clip.register_parameter("logit_scale", nn.Parameter(
            sd.get("logit_scale", torch.tensor(2.6592)).to(self.device, torch.float32).reshape(())))
        # frozen image tower (models_negplus.py:27-28): ViT towers run on the HIP kernels; a ModifiedResNet
        # tower (RN50x4, the argparse default) keeps its weights in the state-dict but cannot encode images
        self.vision = None
        if "visual.proj" in sd and "visual.conv1.weight" in sd:
            vcfg = vision_cfg_from_state_dict(sd)
            self.vision = VisionTower(vcfg["width"], vcfg["layers"], vcfg["heads"], vcfg["patch"], vcfg["res"],
                                      vcfg["embed_dim"], self.device)
            self.vision.load_clip_state_dict(sd)
            for key, view in self.vision.named_views().items():
                p = nn.Parameter(view, requires_grad=self.wo_bank)     # models.py:31-33: frozen unless wo_bank
                _attach(self.clip, "visual." + key, p)
                self._params["visual." + key] = p
        else:
            for k, v in sd.items():
                if k.startswith("visual.") and torch.is_floating_point(v):
                    _attach(self.clip, k.replace("downsample.-1", "downsample.avgpool"), nn.Parameter(v.to(self.device).float(),
                                                                                                   requires_grad=False))
            if "visual.layer1.0.conv1.weight" in sd and "visual.attnpool.positional_embedding" in sd:
                from .resnet_tower import ResNetTower          # ModifiedResNet (RN50x4 = train_negplus.py's default)
                self.vision = ResNetTower(sd, self.device, fast=True)     # bf16 like the ViT tower; exact_eval -> fp32
        self._anchor = torch.zeros((), device=self.device, requires_grad=True)
        self.combining_function = self.element_wise_sum
        self.tau = tau
        self.label_smoothing = label_smoothing
        self.output_dim = cfg["embed_dim"]
        self.input_dim = self._input_resolution(sd)
        self.plus = plus
        self.neg_num = neg_num
        self.tokenizer = tokenizer
        # callable(PIL image | uint8 [H,W,3]) -> fp32 [3, dim, dim] ON THE DEVICE (data_utils.py:84-98 semantics,
        # bit-identical output; use it from the main process, not from DataLoader workers)
        self.preprocess = None
        if transform == "targetpad":
            from .preprocess import TargetPadTransform
            self.preprocess = TargetPadTransform(target_ratio, self.input_dim, self.device)
        self.refer_bank = None
        self._target_bank = None
        self._target_bank_dev = None
        self.M = 0
        self.grad_scale_hook = None     # set by the DDP wrapper

    # -------------------------------------------------------------------------- loading
    @staticmethod
    def _resolve_state_dict(name):
        if isinstance(name, dict):
            return name
        if isinstance(name, str) and name.startswith("synthetic:"):
            from . import synthetic
            w, l, h, d = synthetic.CLIP_TEXT_CONFIGS[name.split(":", 1)[1]]
            return synthetic.text_state_dict(w, l, d)
        path = name
        if isinstance(name, str) and not os.path.isfile(name) and name in _CLIP_FILES:
            # a model NAME: the file clip.load would have downloaded to ~/.cache/clip (clip/clip.py:120-121); there is
            # no network here, so it must already be there (or under $SPN_CLIP_CACHE)
            root = os.environ.get("SPN_CLIP_CACHE", os.path.expanduser("~/.cache/clip"))
            path = os.path.join(root, _CLIP_FILES[name])
        if isinstance(path, str) and os.path.isfile(path):
            # clip/clip.py:127-137: the published files are TorchScript archives; a saved state dict is the fallback
            try:
                sd = torch.jit.load(path, map_location="cpu").state_dict()
            except RuntimeError:
                sd = torch.load(path, map_location="cpu")
            sd = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd
            return {k: v for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        raise RuntimeError(f"Model {name} not found (expected a CLIP file - JIT archive or state dict -, a model name whose "
                           f"file is in ~/.cache/clip, a dict or 'synthetic:<name>')")

    @staticmethod
    def _input_resolution(sd):
        if "visual.conv1.weight" in sd and "visual.positional_embedding" in sd:
            patch = sd["visual.conv1.weight"].shape[-1]
            grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
            return patch * grid
        if "visual.attnpool.positional_embedding" in sd:            # ModifiedResNet: clip/model.py:417-419
            return round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5) * 32
        return 224

    def load_ckpt(self, model_path, is_origin=False):
        """models_negplus.py:52-57: stage-1 `{'CLIP': sd}` when is_origin else `{'state_dict': sd}` (strict=False)."""
        saved = torch.load(model_path, map_location="cpu")
        src = {"clip." + k: v for k, v in saved["CLIP"].items()} if is_origin else saved["state_dict"]
        own = self.state_dict()      # tensors alias the flat parameter buffer
        with torch.no_grad():
            for k, v in src.items():
                if k in own and own[k].shape == v.shape:
                    own[k].copy_(v.to(own[k].device, own[k].dtype))
        self.tower.mark_stale()
        # a stage-1 checkpoint also carries a fine-tuned image tower: re-derive what is computed from its weights
        if isinstance(self.vision, VisionTower):
            self.vision.mark_stale()
        elif self.vision is not None:
            from .resnet_tower import ResNetTower
            vis = {k[len("clip."):]: v for k, v in self.state_dict().items() if k.startswith("clip.visual.")}
            self.vision = ResNetTower({k.replace("downsample.avgpool", "downsample.-1"): v for k, v in vis.items()}, self.device,
                                      fast=True)

    # ---------------------------------------------------------------------------- banks
    @property
    def target_bank(self):
        return self._target_bank

    @target_bank.setter
    def target_bank(self, bank):
        """fp32 [M, D] L2-normalised rows (models_negplus.py:76-77); mirrored on the device as bf16."""
        self._target_bank = bank
        self._target_bank_dev = None if bank is None else ops.prepare_bank(bank.to(self.device, torch.float32))

    def set_target_bank_shard(self, bank_shard_dev, m_begin, m_total):
        """DDP bank-sharded mode: this rank holds rows [m_begin, m_begin + rows) of the global bank."""
        self._target_bank_dev = bank_shard_dev
        self._shard = (m_begin, m_total)

    def load_refer_bank(self, bank_path):
        self.refer_bank = torch.load(bank_path)

    # Bank builders (models_negplus.py:59-125).  `dataset` is duck-typed like the reference's CIRDataset in
    # 'relative' train mode: len(), `.image_id` (number of unique images) and items
    # (reference_image, caption, target_image, index, target_index, reference_index_all, target_index_all)
    # with images already preprocessed to fp32 [3, res, res]; 'unlabeled' mode yields single images.
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
        """models_negplus.py:59-80: per-triplet raw reference features + normalised unique-image target bank."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            self.refer_bank, self.target_bank = torch.load(bank_path)
            return
        refer = torch.zeros(len(cirDataset), self.output_dim)
        target = torch.zeros(cirDataset.image_id, self.output_dim)
        for items in self._image_batches(cirDataset):
            ref = self.encode_image(stack_images([it[0] for it in items]))
            tgt = self.encode_image(stack_images([it[2] for it in items]))
            index = torch.tensor([int(it[3]) for it in items])
            ref_all = torch.tensor([int(it[5]) for it in items])
            tgt_all = torch.tensor([int(it[6]) for it in items])
            refer[index] = ref.cpu()
            target[ref_all] = ops.combine_l2norm_fwd(None, None, ref)[0].cpu()
            target[tgt_all] = ops.combine_l2norm_fwd(None, None, tgt)[0].cpu()
        self.refer_bank, self.target_bank = refer, target
        if bank_path:
            torch.save([refer, target], bank_path)

    def extract_refer_bank_features(self, cirDataset, device=None, bank_path=None, reload_bank=False):
        """models_negplus.py:82-98: raw features per unique image id, kept on the device."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            return
        refer = torch.zeros(cirDataset.image_id, self.output_dim, device=self.device)
        for items in self._image_batches(cirDataset):
            ref_all = torch.tensor([int(it[5]) for it in items], device=self.device)
            tgt_all = torch.tensor([int(it[6]) for it in items], device=self.device)
            refer[ref_all] = self.encode_image(stack_images([it[0] for it in items]))
            refer[tgt_all] = self.encode_image(stack_images([it[2] for it in items]))
        self.refer_bank = refer
        if bank_path:
            torch.save(refer, bank_path)

    def extract_unlabeled_bank_features(self, cirDataset, device=None, bank_path=None, reload_bank=False):
        """models_negplus.py:100-125: normalised features of images outside every triplet, appended after the
        labelled rows (self.M = number of labelled rows); neg_num > 0 keeps the first neg_num of them."""
        if bank_path and os.path.exists(bank_path) and not reload_bank:
            unl = torch.load(bank_path)[0]
        else:
            feats = []
            for items in self._image_batches(cirDataset):
                f = self.encode_image(stack_images(items))
                feats.append(ops.combine_l2norm_fwd(None, None, f)[0].cpu())
            unl = torch.cat(feats) if feats else torch.zeros(0, self.output_dim)
            if bank_path:
                torch.save([unl], bank_path)
        if self.neg_num > 0:
            unl = unl[:self.neg_num, :]
        self.unlabeled_target_bank = unl
        self.M = self.target_bank.shape[0]
        self.target_bank = torch.cat([self.target_bank, unl])

    # -------------------------------------------------------------------------- encoders
    def tokenize(self, text):
        self._pack = (None, 0)
        if torch.is_tensor(text):
            if self.pack_eot and not text.is_cuda:
                self._set_pack(text)
                text = text[:, :self.tower.live_length(text)]         # padding columns only beyond the longest caption
            return text.to(self.device, torch.int32).contiguous()
        if self.tokenizer is None:
            from .tokenizer import tokenize as clip_tokenize      # clip.tokenize (clip/clip.py:206-247)
            self.tokenizer = clip_tokenize
        ids = self.tokenizer(text)
        if int(ids.max()) >= self.tower.vocab:
            raise RuntimeError(f"token id {int(ids.max())} outside the model's vocabulary ({self.tower.vocab})")
        if self.pack_eot:
            self._set_pack(ids)
            ids = ids[:, :self.tower.live_length(ids)]
        return ids.to(self.device, torch.int32).contiguous()

    def _set_pack(self, ids_host):
        """pack_eot: the ids are still on the host here, so the live lengths cost no device sync."""
        cu, total = self.tower.cu_seqlens(ids_host)
        self._pack = (cu.to(self.device), total)

    def encode_image(self, image):
        """fp32 [B, 3, res, res] -> un-normalised image features [B, D] (models_negplus.py:39-41)."""
        if self.vision is None:
            raise RuntimeError("this checkpoint has no image tower")
        with torch.no_grad():
            if self.exact_eval:
                return self.vision.forward_exact(image.to(self.device, torch.float32))
            return self.vision.forward(image)

    def encode_text(self, text):
        """list[str] (or pre-tokenised ids) -> un-normalised text features [B, D] (models_negplus.py:43-46)."""
        ids = self.tokenize(text)
        if self.exact_eval and not torch.is_grad_enabled():
            return self.tower.forward_exact(ids)
        return self.tower.forward(ids, *self._pack)

    def element_wise_sum(self, refer_image_feats, text_feats):
        return refer_image_feats + text_feats        # models_negplus.py:48-50

    element_wise_sum._spn_fused_sum = True           # validate._predict: this Combiner has a fused kernel (spn_combine_l2norm_fwd)

    # ------------------------------------------------------------------------------ step
    def _refer_rows(self, indexs, refer_indexs):
        idx = refer_indexs if self.plus else indexs      # models_negplus.py:132-135
        bank = self.refer_bank
        if bank.device != self.device:
            bank = self.refer_bank = bank.to(self.device)
        return bank, idx.to(self.device, torch.int64)

    def _step_forward(self, ids, refer_idx, labels):
        bank_dev = self._target_bank_dev
        feats = self.tower.forward(ids, *self._pack)
        q, qb, inv = ops.combine_l2norm_fwd(self._refer_f32, refer_idx, feats)
        M = bank_dev.shape[0]
        saved = ops.bank_logits_buffer(qb.shape[0], M, qb.device) if torch.is_grad_enabled() else None
        stats = ops.bank_stats_fwd(qb, bank_dev, labels, 1.0 / self.tau, save=saved)
        lse, row, mean = ops.bank_loss_finalize(stats, M, self.label_smoothing)
        return dict(q=q, qb=qb, inv=inv, lse=lse, loss=mean, labels=labels, B=ids.shape[0], M=M, saved=saved)

    @staticmethod
    def _dev_scale(grad_out, device):
        """autograd's incoming d(loss) as a 1-element fp32 device tensor: it scales the (linear) backward on the device, so
        the backward pass starts without a host synchronisation (the reference's GradScaler hands a scaled loss here)."""
        if torch.is_tensor(grad_out):
            return grad_out.detach().to(device=device, dtype=torch.float32).reshape(1)
        return torch.full((1,), float(grad_out), dtype=torch.float32, device=device)

    def _step_backward(self, st, grad_out):
        dq = ops.bank_grad_q(st["qb"], self._target_bank_dev, st["labels"], 1.0 / self.tau, st["lse"],
                             1.0 / st["B"], M_total=st["M"], label_smoothing=self.label_smoothing, saved=st["saved"])
        dtext = ops.combine_l2norm_bwd(st["q"], st["inv"], dq[:, :self.output_dim].contiguous(),
                                       scale=self._dev_scale(grad_out, self.device))
        snap = gradsink.snapshot(self._params, self.tower.grads, self.tower.named_views)
        flat = self.tower.backward(dtext)
        gradsink.publish(self._params, flat, self.tower.named_views, snap)

    # ---------------------------------------------------------------- in-batch step (config 1)
    def _inbatch_forward(self, ids, refer_image, target_image):
        """clip4cir/models.py:151-167 with wo_bank: both images through the (trainable) visual tower as ONE
        batch of 2B, q = normalize(ref + text), t = normalize(tgt), CE over the B x B logits, labels = arange."""
        if self.vision is None:
            raise RuntimeError("wo_bank needs a ViT image tower")
        B = ids.shape[0]
        imgs = torch.cat([refer_image, target_image]).to(self.device, torch.float32)
        img_feats = self.vision.forward_train(imgs)
        text_feats = self.tower.forward(ids, *self._pack)
        ar = torch.arange(B, device=self.device, dtype=torch.int64)
        q, qb, inv_q = ops.combine_l2norm_fwd(img_feats[:B].contiguous(), ar, text_feats)
        t, tb, inv_t = ops.combine_l2norm_fwd(None, None, img_feats[B:].contiguous())
        stats = ops.bank_stats_fwd(qb, tb, ar, 1.0 / self.tau)
        lse, row, mean = ops.bank_loss_finalize(stats, B, 0.0)
        return dict(q=q, qb=qb, inv_q=inv_q, t=t, tb=tb, inv_t=inv_t, lse=lse, loss=mean, labels=ar, B=B)

    def _inbatch_backward(self, st, grad_out):
        scale = self._dev_scale(grad_out, self.device)
        B, D = st["B"], self.output_dim
        gs = 1.0 / B
        dq = ops.bank_grad_q(st["qb"], st["tb"], st["labels"], 1.0 / self.tau, st["lse"], gs, M_total=B)
        dt = ops.inbatch_grad_t(st["qb"], st["tb"], st["lse"], 1.0 / self.tau, gs, D)
        dsum = ops.combine_l2norm_bwd(st["q"], st["inv_q"], dq[:, :D].contiguous(), scale=scale)   # = d ref_feats = d text_feats
        dtgt = ops.combine_l2norm_bwd(st["t"], st["inv_t"], dt, scale=scale)
        snap_t = gradsink.snapshot(self._params, self.tower.grads, self.tower.named_views)
        snap_v = gradsink.snapshot(self._params, self.vision.grads, self.vision.named_views, "visual.")
        flat_t = self.tower.backward(dsum)
        flat_v = self.vision.backward(torch.cat([dsum, dtgt]))
        gradsink.publish(self._params, flat_t, self.tower.named_views, snap_t)
        gradsink.publish(self._params, flat_v, self.vision.named_views, snap_v, "visual.")

    def forward(self, text, indexs, target_indexs, refer_indexs, refer_image=None, target_image=None):
        """models_negplus.py:144-148 -> {'bank_loss': 0-dim tensor with grad}; with wo_bank
        (clip4cir/models.py:151-160) -> {'bbc_loss': ...} from the two image batches."""
        ids = self.tokenize(text)
        if self.wo_bank:
            loss = _InBatchStep.apply(self._anchor, self, ids, refer_image, target_image)
            return {"bbc_loss": loss}
        ops.check_index_range(refer_indexs if self.plus else indexs, self.refer_bank.shape[0], "refer_bank")
        ops.check_index_range(target_indexs, self.target_bank.shape[0], "target_bank labels")   # CrossEntropyLoss: class < M
        bank, ridx = self._refer_rows(indexs, refer_indexs)
        self._refer_f32 = bank if bank.dtype == torch.float32 else bank.float()
        labels = target_indexs.to(self.device, torch.int64)
        loss = _BankStep.apply(self._anchor, self, ids, ridx, labels)
        return {"bank_loss": loss}

    def parameters_changed(self):
        """Call after an external optimizer updated the parameters in place."""
        self.tower.mark_stale()
        if self.vision is not None and self.wo_bank:
            self.vision.mark_stale()
