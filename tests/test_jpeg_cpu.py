"""Baseline JPEG decode: the numpy oracle (oracle/jpeg_decode.py), the host-side batch builder (spn4cir_amd/jpeg.py) and the exact
per-work-item source of the GPU kernels (csrc/jpeg_core.h, compiled for the host by tests/native/jpeg_core_host.cpp) - all three
against Pillow (libjpeg-turbo) on generated files, bit for bit.  Pillow is what the reference decodes with
(clip4cir/data_utils_negplus.py:17,268-319)."""
import ctypes as C
import io
import os
import subprocess
import tempfile

import numpy as np
import pytest

from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _synth(rng, h, w, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        img = np.stack([xx * 255 // max(1, w - 1), yy * 255 // max(1, h - 1), (xx + yy) * 255 // max(1, w + h - 2)], -1)
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3))
    else:
        base = rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, 3))
        img = np.kron(base, np.ones((8, 8, 1)))[:h, :w] + rng.normal(0, 12, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def _corpus():
    rng = np.random.default_rng(0)
    files = []
    for (h, w) in [(8, 8), (16, 16), (17, 23), (33, 47), (64, 48), (50, 75), (1, 1), (3, 5), (100, 37), (9, 130), (2, 2), (4, 3)]:
        for sub in (0, 1, 2):
            for q, kind in ((35, 2), (75, 0), (95, 1), (100, 1), (90, 2)):
                buf = io.BytesIO()
                Image.fromarray(_synth(rng, h, w, kind)).save(buf, "JPEG", quality=q, subsampling=sub, optimize=(q == 75))
                files.append(buf.getvalue())
    for (h, w) in [(40, 60), (13, 7), (1, 9)]:                                 # grayscale
        buf = io.BytesIO()
        Image.fromarray(_synth(rng, h, w, 2)).convert("L").save(buf, "JPEG", quality=80)
        files.append(buf.getvalue())
    for (h, w, sub, blocks) in [(70, 90, 2, 3), (33, 200, 1, 1), (64, 64, 0, 7), (120, 50, 2, 2)]:    # restart intervals
        buf = io.BytesIO()
        Image.fromarray(_synth(rng, h, w, 2)).save(buf, "JPEG", quality=85, subsampling=sub, restart_marker_blocks=blocks)
        files.append(buf.getvalue())
    buf = io.BytesIO()                                                        # a photo-sized file (FashionIQ-like 400 x 600)
    Image.fromarray(_synth(rng, 600, 400, 2)).save(buf, "JPEG", quality=90)
    files.append(buf.getvalue())
    return files


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


@pytest.fixture(scope="module")
def corpus():
    return _corpus()


def test_numpy_oracle_matches_pillow(corpus):
    from oracle import jpeg_decode
    for k, data in enumerate(corpus[:120] + corpus[-8:-1]):                   # pure-Python Huffman: the big file is left to the C path
        assert np.array_equal(jpeg_decode.decode(data), _pil(data)), k


@pytest.fixture(scope="module")
def host_core():
    d = tempfile.mkdtemp(prefix="spn_jpeg_")
    so = os.path.join(d, "jpeg_core_host.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", so, os.path.join(ROOT, "tests", "native", "jpeg_core_host.cpp")],
                   check=True)
    return C.CDLL(so)


def test_device_source_on_the_host_matches_pillow(corpus, host_core):
    """spn4cir_amd.jpeg.Batch (the arrays the kernels read) + jpeg_core.h's decode_segment / idct_block / pixel_rgb in the kernels'
    order: every file of the corpus, one batch, equals Pillow."""
    from spn4cir_amd import jpeg
    b = jpeg.Batch(corpus)
    assert b.n == len(corpus) and b.segs.shape[0] > b.n                        # restart intervals became extra work items
    coefs = np.empty(b.coef_elems, dtype=np.int16)
    planes = np.zeros(b.plane_bytes, dtype=np.uint8)
    rgb = np.zeros(b.rgb_bytes, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    images, segs, huff, qt, by = (np.ascontiguousarray(x) for x in (b.images, b.segs, b.huff, b.qt, b.bytes))
    rc = host_core.jpeg_core_decode_host(p(by), p(images), b.n, p(segs), segs.shape[0], p(huff), p(qt), p(coefs), C.c_size_t(b.coef_elems),
                                         p(planes), p(rgb))
    assert rc == 0
    for k, data in enumerate(corpus):
        H, W = b.sizes[k]
        got = rgb[b.rgb_off[k]:b.rgb_off[k] + H * W * 3].reshape(H, W, 3)
        assert np.array_equal(got, _pil(data)), (k, H, W)
    # the standard tables are shared: far fewer Huffman records than 4 per file
    assert b.huff.size // jpeg.HUFF_BYTES < 2 * b.n


def test_out_of_scope_files_are_refused():
    from spn4cir_amd import jpeg
    rng = np.random.default_rng(1)
    img = Image.fromarray(_synth(rng, 40, 40, 2))
    cases = {}
    buf = io.BytesIO(); img.save(buf, "JPEG", progressive=True); cases["progressive"] = buf.getvalue()
    buf = io.BytesIO(); img.convert("CMYK").save(buf, "JPEG"); cases["cmyk"] = buf.getvalue()
    buf = io.BytesIO(); img.save(buf, "JPEG", subsampling="4:1:1") if False else img.save(buf, "PNG"); cases["png"] = buf.getvalue()
    cases["truncated"] = cases["progressive"][:30]
    try:                                                                      # RGB stored as is (Adobe transform 0 / no JFIF marker)
        buf = io.BytesIO(); img.save(buf, "JPEG", keep_rgb=True); cases["rgb"] = buf.getvalue()
    except TypeError:
        pass
    for name, data in cases.items():
        with pytest.raises(jpeg.Unsupported):
            jpeg.parse_header(data)


def test_large_photo_chunks_are_split_by_pixel_budget():
    """A decode chunk of the bank builders is 1 024 dataset items = 2 048 images; CIRR / NLVR2 web photos are several megapixels.
    The list is cut into kernel batches by decoded bytes, so neither the 32-bit offsets of the batch description nor the device
    footprint depend on how many files a caller passes (the reference's per-image path has no limit: data_utils_negplus.py:298-304)."""
    from spn4cir_amd import jpeg
    sizes = [(4000, 3000)] * 2048                                  # 36 MB of RGB each: 73.7 GB in one batch would overflow
    runs = jpeg.split_by_pixels(sizes)
    assert [k for r in runs for k in r] == list(range(2048))       # order kept, nothing lost
    per = [sum(3 * sizes[k][0] * sizes[k][1] for k in r) for r in runs]
    assert max(per) <= jpeg.RGB_BUDGET < 2 ** 31 and len(runs) > 64
    # one image larger than the budget still gets a run of its own; small files share one
    runs = jpeg.split_by_pixels([(100, 100), (30000, 20000), (100, 100), (50, 50)], budget=1 << 20)
    assert runs == [[0], [1], [2, 3]]
    assert jpeg.split_by_pixels([]) == []
