"""Host side of the frozen CLIP image tower (VisionTransformer, clip4cir/clip/model.py:206-242).

Stage 2 never trains it (models_negplus.py:27-28): forward() is the inference path of the bank builders and
validation.  forward_train()/backward() serve BASELINE config 1 (clip4cir/models.py:151-167, wo_bank), where the
tower is trainable.  Parameters live in one flat fp32 buffer (layout spn_vision_layout); the bf16 GEMM operands
are derived from it."""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check, lib
from .ops import _p, _stream
from .text_tower import _BLOCK_KEYS


def vision_cfg_from_state_dict(sd, prefix="visual."):
    """clip4cir/clip/model.py:404-411."""
    width = sd[prefix + "conv1.weight"].shape[0]
    patch = sd[prefix + "conv1.weight"].shape[-1]
    grid = round((sd[prefix + "positional_embedding"].shape[0] - 1) ** 0.5)
    layers = len([k for k in sd if k.startswith(prefix) and k.endswith(".attn.in_proj_weight")])
    return dict(width=width, patch=patch, res=patch * grid, layers=layers, heads=width // 64,
                embed_dim=sd[prefix + "proj"].shape[1])


class VisionTower:
    def __init__(self, width, layers, heads, patch, res, embed_dim, device="cuda", kind=0):
        self.kind = kind
        if heads * 64 != width:
            raise ValueError("CLIP ViT towers use head_dim 64 (clip/model.py:262)")
        self.width, self.layers, self.heads = width, layers, heads
        self.patch, self.res, self.embed_dim = patch, res, embed_dim
        self.device = torch.device(device)
        self._lay = _lib.VisionLayout()
        check(lib().spn_vision_layout(C.byref(self._cfg(1)), C.byref(self._lay)), "vision_layout")
        self.params = torch.zeros(int(self._lay.n_params), dtype=torch.float32, device=self.device)
        self.wbf16 = torch.zeros(int(self._lay.n_bf16), dtype=torch.bfloat16, device=self.device)
        self._ws = None
        self._stale = True
        self.grads = None          # allocated by forward_train(): flat fp32, same layout as params
        self._acts = self._bws = None
        self._train_B = 0

    def _cfg(self, B):
        return _lib.VisionCfg(B, self.res, self.patch, self.width, self.heads, self.layers, self.embed_dim, self.kind)

    def spans(self):
        lay, W, D, p = self._lay, self.width, self.embed_dim, self.patch
        if self.kind == 1:
            return self._blip_spans()
        out = [("conv1.weight", lay.conv1, (W, 3, p, p)), ("class_embedding", lay.cls, (W,)),
               ("positional_embedding", lay.pos, (int(lay.seq), W)), ("ln_pre.weight", lay.ln_pre_g, (W,)),
               ("ln_pre.bias", lay.ln_pre_b, (W,))]
        shapes = [(W,), (W,), (3 * W, W), (3 * W,), (W, W), (W,), (W,), (W,), (4 * W, W), (4 * W,), (W, 4 * W), (W,)]
        for l in range(self.layers):
            base = lay.blocks + lay.block_size * l
            for j, key in enumerate(_BLOCK_KEYS):
                out.append((f"transformer.resblocks.{l}.{key}", base + lay.block_off[j], shapes[j]))
        out += [("ln_post.weight", lay.ln_post_g, (W,)), ("ln_post.bias", lay.ln_post_b, (W,)), ("proj", lay.proj, (W, D))]
        return [(k, int(o), s) for k, o, s in out]

    def _blip_spans(self):
        """blip4cir/vit.py state-dict keys (visual_encoder.*) + `vision_proj_t` = vision_proj.weight^T [W, D]."""
        lay, W, D, p = self._lay, self.width, self.embed_dim, self.patch
        out = [("patch_embed.proj.weight", lay.conv1, (W, 3, p, p)), ("patch_embed.proj.bias", lay.conv_b, (W,)),
               ("cls_token", lay.cls, (1, 1, W)), ("pos_embed", lay.pos, (1, int(lay.seq), W))]
        names = ["norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                 "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"]
        shapes = [(W,), (W,), (3 * W, W), (3 * W,), (W, W), (W,), (W,), (W,), (4 * W, W), (4 * W,), (W, 4 * W), (W,)]
        for l in range(self.layers):
            base = lay.blocks + lay.block_size * l
            for j, key in enumerate(names):
                out.append((f"blocks.{l}.{key}", base + lay.block_off[j], shapes[j]))
        out += [("norm.weight", lay.ln_post_g, (W,)), ("norm.bias", lay.ln_post_b, (W,)),
                ("vision_proj_t", lay.proj, (W, D)), ("vision_proj.bias", lay.proj_b, (D,))]
        return [(k, int(o), s) for k, o, s in out]

    def load_blip_state_dict(self, sd, prefix="visual_encoder.", proj_prefix="vision_proj."):
        """BLIP_Retrieval state-dict: visual_encoder.* (vit.py) + vision_proj.{weight [D,W], bias}."""
        with torch.no_grad():
            for key, v in self.named_views().items():
                if key == "vision_proj_t":
                    v.copy_(sd[proj_prefix + "weight"].t().to(self.device, torch.float32))
                elif key == "vision_proj.bias":
                    v.copy_(sd[proj_prefix + "bias"].to(self.device, torch.float32))
                else:
                    v.copy_(sd[prefix + key].to(self.device, torch.float32))
        self._stale = True

    def named_views(self, flat=None):
        flat = self.params if flat is None else flat
        views = {}
        for key, off, shape in self.spans():
            n = 1
            for s in shape:
                n *= s
            views[key] = flat[off:off + n].view(shape)
        return views

    def mark_stale(self):
        self._stale = True

    def is_stale(self):
        """True when the bf16 GEMM operands no longer match the fp32 masters: flagged explicitly (mark_stale) or the
        flat parameter buffer was written in place through any view since the last refresh - torch bumps the shared
        version counter for that, which is how an external `optimizer.step()` on the exposed nn.Parameters
        (train_negplus.py:121-123) is noticed without a parameters_changed() call."""
        return self._stale or self.params._version != getattr(self, "_seen_version", -1)

    def _refresh(self, cfg):
        if self.is_stale():
            check(lib().spn_vision_refresh_bf16(C.byref(cfg), _p(self.params), _p(self.wbf16), _stream()),
                  "vision_refresh_bf16")
            self._stale = False
            self._seen_version = self.params._version

    def forward_exact(self, image):
        """fp32-exact image features of the CLIP tower (see TextTower.forward_exact)."""
        if self.kind != 0:
            raise RuntimeError("the exact mode exists for the CLIP towers only")
        image = image.to(self.device, torch.float32).contiguous()
        B = image.shape[0]
        cfg = self._cfg(B)
        ws = ops.scratch_bytes(lib().spn_vision_exact_ws_bytes(C.byref(cfg)), self.device)
        feats = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        check(lib().spn_vision_fwd_exact(C.byref(cfg), _p(self.params), _p(image), _p(ws), ws.numel(), _p(feats), _stream()),
              "vision_fwd_exact")
        return feats

    # ------------------------------------------------------------------ training (CLIP tower, wo_bank / first stage)
    def forward_train(self, image):
        """As forward(), but keeps every layer's activations for backward() (clip4cir/models.py:156-158 runs the
        tower under torch.utils.checkpoint instead; 288 GB of HBM make the recompute unnecessary)."""
        if self.kind != 0:
            raise RuntimeError("only the CLIP VisionTransformer has a training path")
        image = image.to(self.device, torch.float32).contiguous()
        B = image.shape[0]
        cfg = self._cfg(B)
        self._refresh(cfg)
        if self._train_B != B:
            self._acts = self._bws = None
            self._acts = ops.scratch_bytes(lib().spn_vision_train_act_bytes(C.byref(cfg)), self.device)
            self._bws = ops.scratch_bytes(lib().spn_vision_bwd_ws_bytes(C.byref(cfg)), self.device)
            self._train_B = B
        if self.grads is None:
            self.grads = torch.zeros_like(self.params)
        feats = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        check(lib().spn_vision_fwd_train(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(image), _p(self._acts),
                                         _p(feats), _stream()), "vision_fwd_train")
        return feats

    def backward(self, dfeats):
        """d(loss)/d(feats) fp32 [B, D] of the preceding forward_train() -> self.grads (overwritten)."""
        if not self._train_B or dfeats.shape[0] != self._train_B:
            raise RuntimeError("backward() without a matching forward_train()")
        cfg = self._cfg(self._train_B)
        dfeats = dfeats.to(self.device, torch.float32).contiguous()
        check(lib().spn_vision_bwd(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(self._acts), _p(dfeats),
                                   _p(self.grads), _p(self._bws), self._bws.numel(), _stream()), "vision_bwd")
        return self.grads

    def load_clip_state_dict(self, sd, prefix="visual."):
        with torch.no_grad():
            for key, v in self.named_views().items():
                v.copy_(sd[prefix + key].to(device=self.device, dtype=torch.float32))
        self._stale = True

    def forward(self, image, return_tokens=False):
        """fp32 [B, 3, res, res] (device) -> un-normalised features fp32 [B, D] (and the token sequence [B, S, W]:
        after the final norm for the BLIP ViT, the raw transformer output for CLIP - what TG-CIR's extract_img_fea reads)."""
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != self.res or image.shape[3] != self.res:
            raise ValueError(f"expected [B,3,{self.res},{self.res}], got {tuple(image.shape)}")
        image = image.to(self.device, torch.float32).contiguous()
        B = image.shape[0]
        cfg = self._cfg(B)
        self._refresh(cfg)
        need = lib().spn_vision_ws_bytes(C.byref(cfg))
        if self._ws is None or self._ws.numel() < need:
            self._ws = ops.scratch_bytes(need, self.device)
        feats = torch.empty(B, self.embed_dim, dtype=torch.float32, device=self.device)
        tokens = None
        if return_tokens:
            tokens = torch.empty(B, int(self._lay.seq), self.width, dtype=torch.float32, device=self.device)
        check(lib().spn_vision_fwd(C.byref(cfg), _p(self.params), _p(self.wbf16), _p(image), _p(self._ws),
                                   self._ws.numel(), _p(feats), _p(tokens), _stream()), "vision_fwd")
        return (feats, tokens) if return_tokens else feats
