"""GPU JPEG decode (spn_jpeg_decode_batch through spn4cir_amd.jpeg.decode_batch) against Pillow, bit for bit, and the deferred-decode
path of the image preprocessing (what the bank builders and extract_index_features use)."""
import io
import os
import time

import numpy as np
import pytest
import torch

from PIL import Image

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_decode_batch_matches_pillow_bit_for_bit():
    _need_gpu()
    from test_jpeg_cpu import _corpus, _synth
    from spn4cir_amd import jpeg
    files = _corpus()
    rng = np.random.default_rng(3)
    buf = io.BytesIO()
    Image.fromarray(_synth(rng, 50, 70, 2)).save(buf, "JPEG", progressive=True)        # out of scope: decoded by Pillow on the host
    files.insert(5, buf.getvalue())
    out, fallback = jpeg.decode_batch(files, "cuda")
    assert fallback == [5] and len(out) == len(files)
    for k, (t, data) in enumerate(zip(out, files)):
        want = _pil(data)
        assert t.dtype == torch.uint8 and tuple(t.shape) == want.shape, k
        assert np.array_equal(t.cpu().numpy(), want), k
    # a second call on the same stream reuses nothing of the first (fresh scratch): same bits
    out2, _ = jpeg.decode_batch(files[:9], "cuda")
    assert all(torch.equal(a, b) for a, b in zip(out[:9], out2))
    # a tiny pixel budget cuts the list into many kernel batches (what large-photo datasets hit at the default budget): same bits
    out3, fb3 = jpeg.decode_batch(files, "cuda", rgb_budget=40000)
    assert fb3 == [5] and all(torch.equal(a, b) for a, b in zip(out, out3))


def test_deferred_preprocessing_equals_host_decode(tmp_path):
    """TargetPadTransform(gpu_decode=True): an undecoded PIL JPEG comes back as a DeferredImage (no host decode); stack_images decodes
    the batch on the GPU and preprocesses it - the SAME fp32 tensors as the host-decode path, bit for bit; files out of the kernels'
    scope, PNGs and already-decoded images take the old path."""
    _need_gpu()
    from test_jpeg_cpu import _synth
    from spn4cir_amd.preprocess import DeferredImage, TargetPadTransform, gpu_decode_scope, stack_images
    rng = np.random.default_rng(5)
    paths = []
    for k, (h, w, kw) in enumerate([(300, 200, dict(quality=90)), (120, 400, dict(quality=75, subsampling=1)), (64, 64, dict(quality=95, subsampling=0)),
                                    (500, 333, dict(quality=85, progressive=True)), (90, 90, dict())]):
        p = tmp_path / (f"img{k}.png" if k == 4 else f"img{k}.jpg")
        Image.fromarray(_synth(rng, h, w, 2)).save(p, **({} if k == 4 else dict(format="JPEG", **kw)))
        paths.append(str(p))
    host = TargetPadTransform(1.25, 224, "cuda")
    dev = TargetPadTransform(1.25, 224, "cuda", gpu_decode=True)
    want = [host(Image.open(p)) for p in paths]
    items = [dev(Image.open(p)) for p in paths]
    assert [isinstance(i, DeferredImage) for i in items] == [True, True, True, True, False]     # the PNG is decoded by Pillow as before
    got = stack_images(items)                                                                    # the progressive file falls back inside
    assert tuple(got.shape) == (5, 3, 224, 224)
    for k in range(5):
        assert torch.equal(got[k], want[k]), k
    assert torch.equal(items[0].tensor(), want[0])
    loaded = Image.open(paths[0])
    loaded.load()
    assert torch.is_tensor(dev(loaded))                                                          # already decoded: nothing to defer

    class DS:                                            # the scope the builders use: on inside, off outside
        preprocess = host
    with gpu_decode_scope(DS):
        assert isinstance(DS.preprocess(Image.open(paths[0])), DeferredImage)
    assert torch.is_tensor(DS.preprocess(Image.open(paths[0])))


def test_decode_throughput_report():
    """Not a gate: prints images/s of the batched GPU decode for FashionIQ-sized files next to single-thread Pillow on this host."""
    _need_gpu()
    from test_jpeg_cpu import _synth
    from spn4cir_amd import jpeg
    rng = np.random.default_rng(7)
    files = []
    for k in range(16):
        buf = io.BytesIO()
        Image.fromarray(_synth(rng, 600, 400, 2)).save(buf, "JPEG", quality=90)
        files.append(buf.getvalue())
    files = files * 16                                                                            # 256 files, ~16 distinct
    jpeg.decode_batch(files[:32], "cuda")
    torch.cuda.synchronize()
    for n in (32, 256):
        t0 = time.perf_counter()
        out, fb = jpeg.decode_batch(files[:n], "cuda")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"GPU decode, batch {n}: {n / dt:.0f} images/s ({dt * 1e3:.1f} ms incl. host parsing and upload; {sum(len(f) for f in files[:n]) / n / 1024:.0f} KB per file)")
        assert not fb
    t0 = time.perf_counter()
    for f in files[:32]:
        _pil(f)
    dt = time.perf_counter() - t0
    print(f"Pillow on one host core: {32 / dt:.0f} images/s")
    assert np.array_equal(out[3].cpu().numpy(), _pil(files[3]))


def test_bank_builders_decode_on_the_gpu(golden_dir, tmp_path, monkeypatch):
    """extract_bank_features / extract_index_features over a dataset that does what the reference's CIRDataset does -
    `self.preprocess(PIL.Image.open(path))` per item (data_utils_negplus.py:268-319) - with JPEG files on disk: the builders fetch the
    items with the transform in deferred mode, decode whole chunks on the GPU and produce the SAME banks / index features, bit for bit,
    as with host decoding (SPN_GPU_JPEG=0)."""
    _need_gpu()
    from test_jpeg_cpu import _synth
    from spn4cir_amd import jpeg
    from spn4cir_amd.models import CIRPlus
    from spn4cir_amd.utils import extract_index_features
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    rng = np.random.default_rng(11)
    paths = []
    for k in range(9):
        p = tmp_path / (f"im{k}.png" if k == 4 else f"im{k}.jpg")
        img = Image.fromarray(_synth(rng, 40 + 7 * k, 90 - 5 * k, 2))
        img.save(p, **({} if k == 4 else dict(format="JPEG", quality=80 + k, subsampling=k % 3, progressive=(k == 7))))
        paths.append(str(p))
    model = CIRPlus(sd, device=torch.device("cuda"), plus=True)

    class Train:                                          # relative / train mode items: (ref img, caption, tgt img, idx, tgt id, ref id_all, tgt id_all)
        image_id = 9
        trip = [(0, 1), (2, 3), (4, 5), (6, 7), (8, 0), (7, 2)]
        preprocess = model.preprocess

        def __len__(self):
            return len(self.trip)

        def __getitem__(self, i):
            r, t = self.trip[i]
            return self.preprocess(Image.open(paths[r])), "cap", self.preprocess(Image.open(paths[t])), i, t, r, t

    class Classic:                                        # classic mode items: (name, image)
        preprocess = model.preprocess
        data_name, split, dress_types = "fiq", "val", ["dress"]

        def __len__(self):
            return len(paths)

        def __getitem__(self, i):
            return f"n{i}", self.preprocess(Image.open(paths[i]))

    calls = []
    real = jpeg.decode_batch
    monkeypatch.setattr(jpeg, "decode_batch", lambda files, device="cuda": (calls.append(len(files)), real(files, device))[1])
    model.extract_bank_features(Train(), bank_path=None)
    refer_gpu, target_gpu = model.refer_bank.clone(), model.target_bank.clone()
    feats_gpu, names = extract_index_features(Classic(), model)
    assert calls == [11, 8], calls          # ONE decode call per chunk: 12 image fields minus the one PNG field, 9 files minus the PNG
    monkeypatch.setenv("SPN_GPU_JPEG", "0")
    model.extract_bank_features(Train(), bank_path=None)
    feats_host, names2 = extract_index_features(Classic(), model)
    assert len(calls) == 2 and names == names2 == [f"n{i}" for i in range(9)]
    assert torch.equal(model.refer_bank, refer_gpu) and torch.equal(model.target_bank, target_gpu)
    assert torch.equal(feats_gpu, feats_host)
