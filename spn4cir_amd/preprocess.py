"""Host side of the GPU image preprocessing (targetpad_transform, clip4cir/data_utils.py:42-65,84-98).

The image is decoded on the host (PIL or any decoder) to uint8 RGB and uploaded once; padding, the two-pass
bicubic resampling, centre crop, ToTensor and Normalize run in two HIP kernels (csrc/preprocess.hip).  What the
host computes here is geometry and Pillow's coefficient tables (precompute_coeffs + normalize_coeffs_8bpc,
double arithmetic), cached per (input size, output size)."""
import ctypes as C
import functools
import math

import numpy as np
import torch

from ._lib import check, lib
from .ops import _p, _stream

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=4096)
def _coeffs(in_size, out_size):
    """Pillow's bicubic window for every output coordinate: (int32 [out, ksize], int32 [out, 2] = (first, count))."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def targetpad_geometry(w, h, target_ratio, dim):
    """(pad_x, pad_y, resized_w, resized_h, crop_left, crop_top) of targetpad_transform for a w x h image."""
    hp = vp = 0
    if max(w, h) / min(w, h) >= target_ratio:                    # data_utils.py:58-63
        scaled = max(w, h) / target_ratio
        hp = max(int((scaled - w) / 2), 0)
        vp = max(int((scaled - h) / 2), 0)
    wp, hpad = w + 2 * hp, h + 2 * vp
    if wp <= hpad:                                               # torchvision Resize(int): shorter side -> dim
        ow, oh = dim, int(dim * hpad / wp)
    else:
        ow, oh = int(dim * wp / hpad), dim
    return hp, vp, ow, oh, int(round((ow - dim) / 2.0)), int(round((oh - dim) / 2.0))


class DeferredImage:
    """A JPEG file's bytes standing in for its preprocessed tensor: what `TargetPadTransform(gpu_decode=True)` returns for a
    not-yet-decoded PIL JPEG, so that the callers that handle whole batches (bank builders, extract_index_features) decode the batch
    on the GPU at once (`stack_images`).  `.tensor()` realises one image alone."""
    __slots__ = ("data", "transform")

    def __init__(self, data, transform):
        self.data, self.transform = data, transform

    def tensor(self):
        return self.transform.realize([self])[0]


class gpu_decode_scope:
    """`with gpu_decode_scope(dataset): ...` - while the block runs, a dataset whose `preprocess` is a TargetPadTransform hands out
    DeferredImage for undecoded JPEGs, so that the block's `stack_images` decodes them on the GPU batch by batch.  The callers that
    use it (bank builders, extract_index_features) run in the main process and consume the items themselves; everybody else keeps
    getting tensors from `model.preprocess(img)`.  SPN_GPU_JPEG=0 switches it off (host decode everywhere)."""

    def __init__(self, dataset):
        import os
        t = getattr(dataset, "preprocess", None)
        self.t = t if isinstance(t, TargetPadTransform) and os.environ.get("SPN_GPU_JPEG", "1") != "0" else None

    def __enter__(self):
        if self.t is not None:
            self.old, self.t.gpu_decode = self.t.gpu_decode, True
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            self.t.gpu_decode = self.old
        return False


def realize_items(items):
    """[tuple / list / single] dataset items whose fields may be DeferredImage -> the same items with every deferred field replaced by
    its preprocessed tensor; ALL deferred images of the list go through one batched GPU decode (a lane per file: the more files per
    call, the more of the device decodes - tools/jpeg_bench.py)."""
    refs = []
    for i, it in enumerate(items):
        if isinstance(it, DeferredImage):
            refs.append((i, None, it))
        elif isinstance(it, (tuple, list)):
            refs.extend((i, j, f) for j, f in enumerate(it) if isinstance(f, DeferredImage))
    if not refs:
        return list(items)
    done = refs[0][2].transform.realize([r[2] for r in refs])
    out = [list(it) if isinstance(it, (tuple, list)) else it for it in items]
    for (i, j, _), t in zip(refs, done):
        if j is None:
            out[i] = t
        else:
            out[i][j] = t
    return [tuple(o) if isinstance(items[k], tuple) else o for k, o in enumerate(out)]


def stack_images(images):
    """[n] preprocessed image tensors and / or DeferredImage -> fp32 [n, 3, dim, dim] on the device: the deferred ones are decoded
    in ONE batched GPU call (spn_jpeg_decode_batch) and preprocessed from device memory."""
    images = list(images)
    pending = [i for i, im in enumerate(images) if isinstance(im, DeferredImage)]
    if pending:
        done = images[pending[0]].transform.realize([images[i] for i in pending])
        for i, t in zip(pending, done):
            images[i] = t
    return torch.stack([im.to(images[0].device) for im in images])


class TargetPadTransform:
    """GPU `targetpad_transform(target_ratio, dim)`: uint8 RGB [H, W, 3] (numpy / CPU or device tensor, or a PIL
    image) -> fp32 [3, dim, dim] on the device, bit-identical to the reference's CPU pipeline.  PIL images in modes
    other than RGB / L are padded / resized / cropped by Pillow in their own mode first (`_native_mode_u8`).
    gpu_decode=True: a PIL JPEG that has not been decoded yet (`PIL.Image.open(path)` only reads the header - the reference's
    datasets hand exactly that to the transform, data_utils_negplus.py:268-319) is NOT decoded on the host: its file bytes come back
    as a `DeferredImage`, and `stack_images` / `realize` decode whole batches on the GPU (spn4cir_amd/jpeg.py)."""

    def __init__(self, target_ratio=1.25, dim=224, device="cuda", mean=CLIP_MEAN, std=CLIP_STD, gpu_decode=False):
        self.target_ratio, self.dim, self.device = float(target_ratio), int(dim), torch.device(device)
        self.gpu_decode = bool(gpu_decode)
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)
        self._tables = {}

    def _device_tables(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._tables:
            kk, bounds = _coeffs(in_size, out_size)
            self._tables[key] = (torch.from_numpy(kk).to(self.device), torch.from_numpy(bounds).to(self.device), kk.shape[1])
        return self._tables[key]

    def _native_mode_u8(self, image):
        """PIL image in a mode other than RGB / L -> uint8 RGB [dim, dim, 3] right before ToTensor.

        The reference's Compose pads, resizes and crops in the image's OWN mode and converts to RGB afterwards
        (data_utils.py:91-95).  For RGB that order is immaterial and for L the three channels stay equal, but Pillow
        resamples palettised ('P') and bilevel ('1') images with NEAREST whatever filter is asked for, resizes
        RGBA / LA with premultiplied alpha, and 'I' / 'F' / CMYK have their own arithmetic - convert-first gives
        different pixels there.  Such files are rare (a few PNGs), so they take Pillow's own path on the host and only
        ToTensor + Normalize run on the device (the kernel with identical input and output size is the identity)."""
        from PIL import Image
        w, h = image.size
        hp, vp, ow, oh, left, top = targetpad_geometry(w, h, self.target_ratio, self.dim)
        if hp or vp:                               # torchvision F.pad(img, [hp, vp, hp, vp], 0, 'constant') = ImageOps.expand
            canvas = Image.new(image.mode, (w + 2 * hp, h + 2 * vp), 0)
            if image.mode == "P" and image.palette is not None:
                canvas.putpalette(image.getpalette())
            canvas.paste(image, (hp, vp))
            image = canvas
        if image.size != (ow, oh):
            image = image.resize((ow, oh), Image.BICUBIC)
        image = image.crop((left, top, left + self.dim, top + self.dim))
        return np.array(image.convert("RGB"), dtype=np.uint8)

    def _undecoded_jpeg_bytes(self, image):
        """File bytes of a PIL JPEG whose pixels have not been loaded (None otherwise)."""
        if getattr(image, "format", None) != "JPEG" or getattr(image, "mode", None) not in ("RGB", "L"):
            return None
        d = getattr(image, "__dict__", {})
        if d.get("_im", d.get("im")) is not None:       # already decoded by somebody (Pillow >= 11: _im, before: im): nothing to save
            return None
        name = getattr(image, "filename", "")
        try:
            if name:
                with open(name, "rb") as f:
                    return f.read()
            fp = getattr(image, "fp", None)
            if fp is not None and hasattr(fp, "getvalue"):
                return fp.getvalue()
        except OSError:
            return None
        return None

    def realize(self, deferred):
        """[DeferredImage] -> [fp32 [3, dim, dim]]: one batched GPU decode, then the preprocessing kernels per image."""
        from . import jpeg
        rgb, _ = jpeg.decode_batch([d.data for d in deferred], self.device)
        return [self(t) for t in rgb]

    def __call__(self, image, return_uint8=False):
        if not torch.is_tensor(image):
            if self.gpu_decode and not return_uint8 and hasattr(image, "convert"):
                data = self._undecoded_jpeg_bytes(image)
                if data is not None:
                    return DeferredImage(data, self)
            if hasattr(image, "convert"):
                if image.mode not in ("RGB", "L"):
                    return self(self._native_mode_u8(image), return_uint8)
                image = image.convert("RGB")
            image = torch.from_numpy(np.array(image, dtype=np.uint8))
        if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
            raise ValueError("expected a uint8 RGB image [H, W, 3]")
        src = image.to(self.device).contiguous()
        H, W = int(src.shape[0]), int(src.shape[1])
        hp, vp, ow, oh, left, top = targetpad_geometry(W, H, self.target_ratio, self.dim)
        kx, bx, ksx = self._device_tables(W + 2 * hp, ow)
        ky, by, ksy = self._device_tables(H + 2 * vp, oh)
        d = self.dim
        tmp = torch.empty((H + 2 * vp) * d * 3, dtype=torch.uint8, device=self.device)
        out = torch.empty(3, d, d, dtype=torch.float32, device=self.device)
        u8 = torch.empty(d, d, 3, dtype=torch.uint8, device=self.device) if return_uint8 else None
        check(lib().spn_preprocess_image(_p(src), H, W, hp, vp, _p(kx), _p(bx), ksx, _p(ky), _p(by), ksy, left, top, d,
                                         self._mean, self._std, _p(tmp), _p(out), _p(u8), _stream()), "preprocess_image")
        return (out, u8) if return_uint8 else out
