"""Host side of CLIP's ModifiedResNet image towers (RN50 / RN101 / RN50x4..., clip4cir/clip/model.py:10-155) for
`encode_image` (bank builders, validation): frozen, inference only; fp32 (exact, default) or bf16 (`fast=True`).

Activations are NHWC fp32 on the device; every convolution is `spn_im2col3x3_f32` (3x3) or nothing (1x1) followed by
`spn_gemm_f32` with the eval-mode BatchNorm folded into weight and bias once at load time; AvgPool2d and the
attention pool have their own small kernels.  The layer sequence is driven from here (about 3 launches per
convolution); the arithmetic is all in csrc/exact.hip.

`fast=True`: NHWC bf16 activations with channels padded to a multiple of 64, convolutions on the bf16 MFMA GEMM
(`spn_im2col3x3_nhwc_bf16` / `spn_im2col3x3_stem_bf16` + `spn_gemm_nt`), ReLU / residual add and the pools in
csrc/resnet.hip; the attention pool at the end stays fp32.  Throughput mode for bank extraction and validation."""
import ctypes as C

import torch

from . import ops as _ops
from ._lib import check, lib
from .ops import _p, _stream

ACT_NONE, ACT_RELU_POST = 0, 3


def resnet_cfg_from_state_dict(sd, prefix="visual."):
    """clip/model.py:412-419."""
    counts = [len({k.split(".")[2] for k in sd if k.startswith(f"{prefix}layer{b}.")}) for b in (1, 2, 3, 4)]
    width = sd[prefix + "layer1.0.conv1.weight"].shape[0]
    grid = round((sd[prefix + "attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
    return dict(layers=tuple(counts), width=width, res=grid * 32, heads=width * 32 // 64,
                embed_dim=sd[prefix + "attnpool.c_proj.weight"].shape[0])


def _fold(conv_w, bn, eps=1e-5):
    """conv [Co, Ci, kh, kw] + eval BatchNorm -> ([Co, K padded to a multiple of 4] fp32, bias [Co])."""
    s = bn["weight"].float() / torch.sqrt(bn["running_var"].float() + eps)
    w = (conv_w.float() * s[:, None, None, None]).reshape(conv_w.shape[0], -1)
    b = bn["bias"].float() - bn["running_mean"].float() * s
    K = w.shape[1]
    Kp = (K + 3) // 4 * 4
    if Kp != K:
        w = torch.cat([w, torch.zeros(w.shape[0], Kp - K)], dim=1)
    return w.contiguous(), b.contiguous()


def _pad64(c):
    return (c + 63) // 64 * 64


def _fold_fast(conv_w, bn, eps=1e-5):
    """conv [Co, Ci, kh, kw] + eval BatchNorm -> (bf16 [Co_p, kh*kw*Ci_p] in tap-major / channel-minor column order,
    fp32 bias [Co_p]); Ci = 3 (the stem's first convolution) is packed as [Co_p, 64] with column = tap * 3 + channel."""
    s = bn["weight"].float() / torch.sqrt(bn["running_var"].float() + eps)
    w = conv_w.float() * s[:, None, None, None]
    b = bn["bias"].float() - bn["running_mean"].float() * s
    Co, Ci, kh, kw = w.shape
    Cop = _pad64(Co)
    w = w.permute(0, 2, 3, 1)                                   # [Co, kh, kw, Ci]
    if Ci == 3:
        w = w.reshape(Co, kh * kw * 3)
        w = torch.cat([w, torch.zeros(Co, 64 - w.shape[1])], dim=1)
    else:
        Cip = _pad64(Ci)
        w = torch.cat([w, torch.zeros(Co, kh, kw, Cip - Ci)], dim=3).reshape(Co, kh * kw * Cip)
    w = torch.cat([w, torch.zeros(Cop - Co, w.shape[1])], dim=0)
    b = torch.cat([b, torch.zeros(Cop - Co)])
    return w.to(torch.bfloat16).contiguous(), b.contiguous()


class ResNetTower:
    def __init__(self, sd, device="cuda", prefix="visual.", fast=False):
        self.device = torch.device(device)
        self.fast = bool(fast)
        cfg = resnet_cfg_from_state_dict(sd, prefix)
        self.layers, self.width, self.res = cfg["layers"], cfg["width"], cfg["res"]
        self.heads, self.embed_dim = cfg["heads"], cfg["embed_dim"]
        g = {k[len(prefix):]: v.detach().cpu() for k, v in sd.items() if k.startswith(prefix)}   # folding runs on the host
        bn = lambda p: {n: g[p + n] for n in ("weight", "bias", "running_mean", "running_var")}
        dev = lambda t: t.to(self.device)
        self.stem = [tuple(map(dev, _fold(g[f"conv{i}.weight"], bn(f"bn{i}.")))) for i in (1, 2, 3)]
        self.blocks = []
        for li, n in enumerate(self.layers, start=1):
            for bi in range(n):
                p = f"layer{li}.{bi}."
                blk = dict(stride=2 if (bi == 0 and li > 1) else 1,
                           c1=tuple(map(dev, _fold(g[p + "conv1.weight"], bn(p + "bn1.")))),
                           c2=tuple(map(dev, _fold(g[p + "conv2.weight"], bn(p + "bn2.")))),
                           c3=tuple(map(dev, _fold(g[p + "conv3.weight"], bn(p + "bn3.")))), down=None)
                if p + "downsample.0.weight" in g:
                    blk["down"] = tuple(map(dev, _fold(g[p + "downsample.0.weight"], bn(p + "downsample.1."))))
                self.blocks.append(blk)
        if self.fast:
            ff = lambda cw, b_: tuple(map(dev, _fold_fast(g[cw], bn(b_))))
            self.fstem = [ff(f"conv{i}.weight", f"bn{i}.") for i in (1, 2, 3)]
            self.fblocks = []
            for li, n in enumerate(self.layers, start=1):
                for bi in range(n):
                    p = f"layer{li}.{bi}."
                    fb = dict(c1=ff(p + "conv1.weight", p + "bn1."), c2=ff(p + "conv2.weight", p + "bn2."),
                              c3=ff(p + "conv3.weight", p + "bn3."), down=None)
                    if p + "downsample.0.weight" in g:
                        fb["down"] = ff(p + "downsample.0.weight", p + "downsample.1.")
                    self.fblocks.append(fb)
        a = "attnpool."
        self.pos = dev(g[a + "positional_embedding"].float().contiguous())
        self.proj = {n: (dev(g[a + n + "_proj.weight"].float().contiguous()), dev(g[a + n + "_proj.bias"].float().contiguous()))
                     for n in ("q", "k", "v", "c")}

    # ------------------------------------------------------------------ primitives
    def _gemm(self, a, w, bias, M, N, K, lda, act=ACT_NONE, resid=None):
        out = torch.empty(M, N, dtype=torch.float32, device=self.device)
        check(lib().spn_gemm_f32(_p(a), _p(w), M, N, K, lda, w.shape[1], 0, _p(bias), act, _p(resid), N, _p(out), N, 1.0,
                                 _stream()), "gemm_f32")
        return out

    def _conv3x3(self, x, B, H, W, Cin, wb, stride=1, nchw=False, act=ACT_RELU_POST):
        w, b = wb
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        ldk = w.shape[1]
        cols = torch.empty(B * Ho * Wo, ldk, dtype=torch.float32, device=self.device)
        check(lib().spn_im2col3x3_f32(_p(x), _p(cols), B, H, W, Cin, stride, int(nchw), ldk, _stream()), "im2col3x3")
        return self._gemm(cols, w, b, B * Ho * Wo, w.shape[0], ldk, ldk, act), Ho, Wo

    def _conv1x1(self, x, M, Cin, wb, act=ACT_NONE, resid=None):
        w, b = wb
        return self._gemm(x, w, b, M, w.shape[0], Cin, Cin, act, resid)

    def _avgpool(self, x, B, H, W, Cc, k):
        y = torch.empty(B * (H // k) * (W // k), Cc, dtype=torch.float32, device=self.device)
        check(lib().spn_avgpool_nhwc_f32(_p(x), _p(y), B, H, W, Cc, k, _stream()), "avgpool")
        return y, H // k, W // k

    # ------------------------------------------------------------------ bf16 fast path
    def _f_relu(self, y, resid=None):
        check(lib().spn_relu_add_bf16(_p(y), _p(resid), y.numel(), _stream()), "relu_add_bf16")
        return y

    def _f_conv3x3(self, x, B, H, W, wb, stride=1):
        w, b = wb
        Cp = w.shape[1] // 9
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        cols = torch.empty(B * Ho * Wo, 9 * Cp, dtype=torch.bfloat16, device=self.device)
        check(lib().spn_im2col3x3_nhwc_bf16(_p(x), _p(cols), B, H, W, Cp, stride, _stream()), "im2col3x3_nhwc_bf16")
        return self._f_relu(_ops.gemm_nt(cols, w, b)), Ho, Wo

    def _f_avgpool(self, x, B, H, W, k):
        Cp = x.shape[1]
        y = torch.empty(B * (H // k) * (W // k), Cp, dtype=torch.bfloat16, device=self.device)
        check(lib().spn_avgpool_nhwc_bf16(_p(x), _p(y), B, H, W, Cp, k, _stream()), "avgpool_nhwc_bf16")
        return y, H // k, W // k

    def _forward_fast_trunk(self, image):
        """image fp32 [B, 3, res, res] -> (x fp32 [B * H * W, C] NHWC feature map of layer4, B, H, W, C)."""
        B, H, W = image.shape[0], self.res, self.res
        w, b = self.fstem[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        cols = torch.empty(B * Ho * Wo, 64, dtype=torch.bfloat16, device=self.device)
        check(lib().spn_im2col3x3_stem_bf16(_p(image), _p(cols), B, H, W, 2, _stream()), "im2col3x3_stem_bf16")
        x = self._f_relu(_ops.gemm_nt(cols, w, b))
        H, W = Ho, Wo
        x, H, W = self._f_conv3x3(x, B, H, W, self.fstem[1])
        x, H, W = self._f_conv3x3(x, B, H, W, self.fstem[2])
        x, H, W = self._f_avgpool(x, B, H, W, 2)
        for blk, fb in zip(self.blocks, self.fblocks):
            out = self._f_relu(_ops.gemm_nt(x, *fb["c1"]))
            out, _, _ = self._f_conv3x3(out, B, H, W, fb["c2"])
            Ho, Wo = H, W
            identity = x
            if blk["stride"] > 1:
                out, Ho, Wo = self._f_avgpool(out, B, H, W, blk["stride"])
            if fb["down"] is not None:
                if blk["stride"] > 1:
                    identity, _, _ = self._f_avgpool(x, B, H, W, blk["stride"])
                identity = _ops.gemm_nt(identity, *fb["down"])
            x = self._f_relu(_ops.gemm_nt(out, *fb["c3"]), identity)
            H, W = Ho, Wo
        Cc = self.blocks[-1]["c3"][0].shape[0]
        return x[:, :Cc].float().contiguous(), B, H, W, Cc

    # ------------------------------------------------------------------ forward
    def forward(self, image):
        """fp32 [B, 3, res, res] -> un-normalised image features fp32 [B, embed_dim] (clip/model.py:139-154)."""
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != self.res or image.shape[3] != self.res:
            raise ValueError(f"expected [B,3,{self.res},{self.res}], got {tuple(image.shape)}")
        x = image.to(self.device, torch.float32).contiguous()
        B, H, W = x.shape[0], self.res, self.res
        if self.fast:
            # the GEMM addresses its operands with 32-bit byte offsets: keep the largest im2col (the stem's 3x3
            # convolutions at res / 2) under 4 GB by running the batch in slices
            per_image = (self.res // 2) ** 2 * 9 * _pad64(self.width) * 2
            step = max(1, min(B, (2 ** 32 - 1) // per_image))
            outs = []
            for b0 in range(0, B, step):
                xs, b, H, W, Cc = self._forward_fast_trunk(x[b0:b0 + step].contiguous())
                outs.append(self._attnpool(xs, b, H, W, Cc))
            return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        x, H, W = self._conv3x3(x, B, H, W, 3, self.stem[0], stride=2, nchw=True)
        x, H, W = self._conv3x3(x, B, H, W, self.width // 2, self.stem[1])
        x, H, W = self._conv3x3(x, B, H, W, self.width // 2, self.stem[2])
        x, H, W = self._avgpool(x, B, H, W, self.width, 2)
        Cc = self.width
        for blk in self.blocks:
            planes = blk["c1"][0].shape[0]
            out = self._conv1x1(x, B * H * W, Cc, blk["c1"], ACT_RELU_POST)
            out, _, _ = self._conv3x3(out, B, H, W, planes, blk["c2"])
            Ho, Wo = H, W
            identity = x
            if blk["stride"] > 1:
                out, Ho, Wo = self._avgpool(out, B, H, W, planes, blk["stride"])
            if blk["down"] is not None:
                if blk["stride"] > 1:
                    identity, _, _ = self._avgpool(x, B, H, W, Cc, blk["stride"])
                identity = self._conv1x1(identity, B * Ho * Wo, Cc, blk["down"])
            x = self._conv1x1(out, B * Ho * Wo, planes, blk["c3"], ACT_RELU_POST, resid=identity)
            H, W, Cc = Ho, Wo, planes * 4
        return self._attnpool(x, B, H, W, Cc)

    def _attnpool(self, x, B, H, W, Cc):
        """AttentionPool2d on the fp32 NHWC feature map x [B * H * W, Cc] (clip/model.py:58-91)."""
        S = H * W + 1
        tok = torch.empty(B * S, Cc, dtype=torch.float32, device=self.device)
        check(lib().spn_attnpool_tokens_f32(_p(x), _p(self.pos), _p(tok), B, H * W, Cc, _stream()), "attnpool_tokens")
        if self.fast and Cc % 64 == 0:
            # throughput mode: the four projections on the bf16 MFMA GEMM (fp32 accumulation and outputs), the attention
            # itself stays fp32
            if not hasattr(self, "_proj_bf16"):
                self._proj_bf16 = {n: w.to(torch.bfloat16).contiguous() for n, (w, _) in self.proj.items()}
            tb = _ops.cast_bf16(tok)
            q = _ops.gemm_nt(tb.view(B, S, Cc)[:, 0].contiguous(), self._proj_bf16["q"], self.proj["q"][1], out_dtype=torch.float32)
            k = _ops.gemm_nt(tb, self._proj_bf16["k"], self.proj["k"][1], out_dtype=torch.float32)
            v = _ops.gemm_nt(tb, self._proj_bf16["v"], self.proj["v"][1], out_dtype=torch.float32)
            o = torch.empty(B, Cc, dtype=torch.float32, device=self.device)
            check(lib().spn_attnpool_attend_f32(_p(q), _p(k), _p(v), _p(o), B, S, Cc // 64, _stream()), "attnpool_attend")
            if self.embed_dim % 8 == 0:
                return _ops.gemm_nt(_ops.cast_bf16(o), self._proj_bf16["c"], self.proj["c"][1], out_dtype=torch.float32)
            return self._gemm(o, self.proj["c"][0], self.proj["c"][1], B, self.embed_dim, Cc, Cc)
        q = self._gemm(tok, self.proj["q"][0], self.proj["q"][1], B, Cc, Cc, S * Cc)          # rows b*S: the pooled token
        k = self._gemm(tok, self.proj["k"][0], self.proj["k"][1], B * S, Cc, Cc, Cc)
        v = self._gemm(tok, self.proj["v"][0], self.proj["v"][1], B * S, Cc, Cc, Cc)
        o = torch.empty(B, Cc, dtype=torch.float32, device=self.device)
        check(lib().spn_attnpool_attend_f32(_p(q), _p(k), _p(v), _p(o), B, S, Cc // 64, _stream()), "attnpool_attend")
        return self._gemm(o, self.proj["c"][0], self.proj["c"][1], B, self.embed_dim, Cc, Cc)

    def forward_exact(self, image):
        """The fp32 path regardless of `fast`."""
        fast, self.fast = self.fast, False
        try:
            return self.forward(image)
        finally:
            self.fast = fast
