"""CPU oracle of the reference's image preprocessing (test infrastructure only).

clip4cir/data_utils.py:84-98  targetpad_transform = Compose([TargetPad(ratio, dim), Resize(dim, BICUBIC),
CenterCrop(dim), _convert_image_to_rgb, ToTensor(), Normalize(CLIP mean, std)]).

TargetPad (data_utils.py:42-65) is restated below and pinned to padding vectors captured from the reference
class (tests/golden/targetpad.npz).  Resize / CenterCrop / ToTensor / Normalize are torchvision transforms
(requirements.txt: torchvision, unpinned; not installed in this image), restated from their published
definitions: Resize(int) scales the SHORTER side to `dim` and the longer one to int(dim * long / short);
CenterCrop offsets are int(round((size - dim) / 2.0)) (Python round); ToTensor = uint8 HWC -> fp32 CHW / 255;
Normalize = (x - mean) / std.  The resampling itself is done by Pillow (present in the image), the library
torchvision delegates to."""
import numpy as np
import torch
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def targetpad_padding(w, h, target_ratio):
    """data_utils.py:56-64 -> (hp, vp), both 0 when the image is returned unchanged."""
    actual_ratio = max(w, h) / min(w, h)
    if actual_ratio < target_ratio:
        return 0, 0
    scaled_max_wh = max(w, h) / target_ratio
    hp = max(int((scaled_max_wh - w) / 2), 0)
    vp = max(int((scaled_max_wh - h) / 2), 0)
    return hp, vp


def resized_size(w, h, dim):
    if w <= h:
        return dim, int(dim * h / w)
    return int(dim * w / h), dim


def center_crop_offsets(w, h, dim):
    return int(round((w - dim) / 2.0)), int(round((h - dim) / 2.0))


def targetpad_transform_u8(img, target_ratio, dim):
    """PIL image (any mode) -> uint8 [dim, dim, 3] right before ToTensor.  Pad, resize and crop run in the image's own
    mode, the RGB conversion comes after them (data_utils.py:91-95)."""
    w, h = img.size
    hp, vp = targetpad_padding(w, h, target_ratio)
    if hp or vp:
        # F.pad(image, [hp, vp, hp, vp], 0, 'constant') on a PIL image = ImageOps.expand(border, fill=0); torchvision
        # re-attaches the palette of a 'P' image afterwards
        canvas = Image.new(img.mode, (w + 2 * hp, h + 2 * vp), 0)
        if img.mode == "P" and img.palette is not None:
            canvas.putpalette(img.getpalette())
        canvas.paste(img, (hp, vp))
        img = canvas
    w, h = img.size
    ow, oh = resized_size(w, h, dim)
    if (ow, oh) != (w, h):
        img = img.resize((ow, oh), Image.BICUBIC)
    left, top = center_crop_offsets(ow, oh, dim)
    img = img.crop((left, top, left + dim, top + dim))
    return np.asarray(img.convert("RGB"))


def targetpad_transform(img, target_ratio, dim):
    """-> fp32 [3, dim, dim], as the reference's Compose returns it."""
    u8 = torch.from_numpy(targetpad_transform_u8(img, target_ratio, dim).copy())
    x = u8.permute(2, 0, 1).to(torch.float32).div(255)
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(3, 1, 1)
    return x.sub(mean).div(std)
