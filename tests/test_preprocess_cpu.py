"""Host logic of the image preprocessing (no GPU): the oracle's TargetPad arithmetic against vectors captured
from the reference class, and the host-computed geometry + Pillow coefficient tables of spn4cir_amd.preprocess,
replayed in numpy exactly as csrc/preprocess.hip applies them, against Pillow itself (bit-exact)."""
import os

import numpy as np
import pytest
from PIL import Image


def test_targetpad_padding_matches_reference(golden_dir):
    from oracle import preprocess as op
    from spn4cir_amd import preprocess as hp
    rows = np.load(os.path.join(golden_dir, "targetpad.npz"))["rows"]
    assert len(rows) == 636
    for ratio, w, h, px, py, padded in rows:
        w, h = int(w), int(h)
        assert op.targetpad_padding(w, h, float(ratio)) == (int(px), int(py))
        assert hp.targetpad_geometry(w, h, float(ratio), 224)[:2] == (int(px), int(py))


def _replay(src, target_ratio, dim):
    """numpy replay of resize_h_kernel + resize_v_norm_kernel (uint8 stage) from the host tables."""
    from spn4cir_amd.preprocess import _coeffs, targetpad_geometry
    H, W, _ = src.shape
    px, py, ow, oh, left, top = targetpad_geometry(W, H, target_ratio, dim)
    kx, bx = _coeffs(W + 2 * px, ow)
    ky, by = _coeffs(H + 2 * py, oh)
    padded = np.zeros((H + 2 * py, W + 2 * px, 3), dtype=np.int64)
    padded[py:py + H, px:px + W] = src
    tmp = np.zeros((H + 2 * py, dim, 3), dtype=np.int64)
    for x in range(dim):
        x0, n = bx[x + left]
        acc = (padded[:, x0:x0 + n, :] * kx[x + left, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << 21)
        tmp[:, x] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((dim, dim, 3), dtype=np.uint8)
    for y in range(dim):
        y0, n = by[y + top]
        acc = (tmp[y0:y0 + n] * ky[y + top, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << 21)
        out[y] = np.clip(acc >> 22, 0, 255)
    return out


@pytest.mark.parametrize("w,h", [(300, 200), (200, 300), (640, 480), (333, 1000), (1000, 333), (224, 224), (64, 48),
                                 (125, 100), (517, 389)])
def test_host_tables_reproduce_pillow(w, h):
    from oracle import preprocess as op
    rng = np.random.default_rng(w * 7 + h)
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = op.targetpad_transform_u8(Image.fromarray(src), 1.25, 224)
    assert np.array_equal(_replay(src, 1.25, 224), ref)


def test_native_mode_order_matters_and_host_fallback_matches_oracle():
    """Why TargetPadTransform keeps Pillow's own path for non-RGB modes: for a palettised image, convert-then-resize
    (bicubic on RGB) differs from the reference's resize-then-convert (NEAREST on indices).  The host half of the
    fallback (`_native_mode_u8`) is checked against the oracle here; the device half in test_preprocess_gpu.py."""
    import numpy as np
    from PIL import Image
    from oracle import preprocess as op
    from spn4cir_amd.preprocess import TargetPadTransform
    rng = np.random.default_rng(3)
    rgb = Image.fromarray(rng.integers(0, 256, (150, 310, 3), dtype=np.uint8))
    tf = TargetPadTransform(1.25, 224, device="cpu")
    for img in (rgb.quantize(colors=32), rgb.convert("RGBA"), rgb.convert("1")):
        ref = op.targetpad_transform_u8(img, 1.25, 224)
        assert np.array_equal(tf._native_mode_u8(img), ref)
    pal = rgb.quantize(colors=32)
    assert not np.array_equal(op.targetpad_transform_u8(pal, 1.25, 224), op.targetpad_transform_u8(pal.convert("RGB"), 1.25, 224))
