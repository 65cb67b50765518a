"""GPU parity of the image preprocessing kernels: bit-exact (uint8 crop and fp32 normalised tensor) against the
oracle = Pillow + the restated torchvision glue (oracle/preprocess.py)."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,ratio,dim", [(300, 200, 1.25, 224), (200, 300, 1.25, 224), (640, 480, 1.25, 224),
                                           (333, 1000, 1.25, 224), (1000, 333, 1.25, 224), (224, 224, 1.25, 224),
                                           (64, 48, 1.25, 224), (1500, 1499, 1.25, 224), (517, 389, 2.0, 288),
                                           (97, 803, 1.0, 384)])
def test_targetpad_transform_bit_exact(w, h, ratio, dim):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import preprocess as op
    from spn4cir_amd.preprocess import TargetPadTransform
    rng = np.random.default_rng(w * 11 + h)
    # smooth content + noise, so the resampler sees both gradients and saturation
    yy, xx = np.mgrid[0:h, 0:w]
    base = (127 + 120 * np.sin(xx / 17.0)[..., None] * np.cos(yy / 23.0)[..., None] * np.array([1, -1, 0.5])).clip(0, 255)
    src = (base + rng.integers(-40, 41, (h, w, 3))).clip(0, 255).astype(np.uint8)
    img = Image.fromarray(src)
    tf = TargetPadTransform(ratio, dim)
    out, u8 = tf(src, return_uint8=True)
    assert np.array_equal(u8.cpu().numpy(), op.targetpad_transform_u8(img, ratio, dim))
    assert torch.equal(out.cpu(), op.targetpad_transform(img, ratio, dim))
    assert torch.equal(tf(img).cpu(), out.cpu())          # PIL input path


@pytest.mark.parametrize("mode", ["P", "RGBA", "LA", "1", "L"])
def test_targetpad_transform_non_rgb_modes(mode):
    """Palettised / alpha / bilevel PNG-style inputs: the reference pads, resizes and crops in the image's own mode and
    converts to RGB afterwards (data_utils.py:91-95) - Pillow resamples 'P' and '1' with NEAREST and premultiplies alpha."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import preprocess as op
    from spn4cir_amd.preprocess import TargetPadTransform
    rng = np.random.default_rng(7)
    w, h = 410, 260                                       # ratio 1.58 >= 1.25: padded
    yy, xx = np.mgrid[0:h, 0:w]
    base = (127 + 120 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 13.0)[..., None] * np.array([1, -1, 0.5])).clip(0, 255)
    rgb = Image.fromarray((base + rng.integers(-30, 31, (h, w, 3))).clip(0, 255).astype(np.uint8))
    if mode == "P":
        img = rgb.quantize(colors=64)
    elif mode in ("RGBA", "LA"):
        alpha = Image.fromarray((255 * (xx + yy) / (w + h)).astype(np.uint8))
        img = rgb.convert(mode[:-1])
        img.putalpha(alpha)
    else:
        img = rgb.convert(mode)
    assert img.mode == mode
    tf = TargetPadTransform(1.25, 224)
    out, u8 = tf(img, return_uint8=True)
    assert np.array_equal(u8.cpu().numpy(), op.targetpad_transform_u8(img, 1.25, 224))
    assert torch.equal(out.cpu(), op.targetpad_transform(img, 1.25, 224))
