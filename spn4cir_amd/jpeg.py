"""Batched baseline-JPEG decode on the GPU (csrc/jpeg.hip) - host side: marker parsing, table building, batch assembly.

The reference decodes every image on the host: `PIL.Image.open(path)` in the dataset, `.convert("RGB")` as the first step of its
transform (clip4cir/data_utils_negplus.py:17,268-319).  For the calls that decode whole galleries - bank extraction
(models_negplus.py:59-125) and `extract_index_features` (utils.py:24-50) - that host decode is what bounds the frozen image
tower here (ViT-L/14: ~4.4 k images/s on the GPU).  `decode_batch(files)` takes the FILE BYTES of a batch and returns RGB uint8
[H, W, 3] device tensors, bit for bit what Pillow (libjpeg-turbo, default settings) produces; `spn_preprocess_image` consumes them
where they are.

Host work per file: one pass over the markers (a few hundred bytes of tables), done here; Huffman decoding, inverse DCT,
upsampling and colour conversion run on the device.  Scope (everything else raises `Unsupported`, and `decode_batch` then decodes
that file with Pillow on the host): 8-bit baseline / extended-sequential Huffman JPEG, one interleaved scan, grayscale or YCbCr with
1x1 chroma and 1x1 / 2x1 / 2x2 luma sampling.  Restart intervals become independent work items (one lane each)."""

import numpy as np
import torch

IMAGE_WORDS = 40
HUFF_BYTES = 1424

class Unsupported(ValueError):
    """The file is not in the kernels' scope (progressive, CMYK, unusual sampling, ...): decode it with Pillow."""


def _be16(b, i):
    return (b[i] << 8) | b[i + 1]


def parse_header(data):
    """One pass over the markers of a JPEG file -> dict(width, height, comps=[(id, h, v, tq, td, ta)], qt={id: uint16[64] in zigzag
    order}, dc / ac = {id: (counts bytes[16], symbols bytes)}, restart_interval, scan_off).  Raises Unsupported."""
    b = memoryview(data)
    n = len(b)
    if n < 4 or b[0] != 0xFF or b[1] != 0xD8:
        raise Unsupported("no SOI marker")
    pos, qt, dc, ac, ri, frame = 2, {}, {}, {}, 0, None
    saw_jfif, adobe_transform = False, None
    while True:
        if pos + 4 > n or b[pos] != 0xFF:
            raise Unsupported("marker expected")
        while pos < n and b[pos] == 0xFF:
            pos += 1
        m = b[pos]
        pos += 1
        if m == 0x01 or 0xD0 <= m <= 0xD8:
            continue
        if m == 0xD9:
            raise Unsupported("EOI before a scan")
        L = _be16(b, pos)
        s, e = pos + 2, pos + L
        if e > n:
            raise Unsupported("truncated segment")
        if m == 0xDB:
            while s < e:
                pq, tq = b[s] >> 4, b[s] & 15
                s += 1
                if pq:
                    vals = np.frombuffer(b[s:s + 128], dtype=">u2").astype(np.uint16)
                    s += 128
                else:
                    vals = np.frombuffer(b[s:s + 64], dtype=np.uint8).astype(np.uint16)
                    s += 64
                qt[tq] = vals                     # kept in zigzag order, as coded: so are the kernels' coefficient blocks
        elif m == 0xC0 or m == 0xC1:
            if b[s] != 8:
                raise Unsupported("sample precision other than 8 bits")
            nf = b[s + 5]
            frame = dict(height=_be16(b, s + 1), width=_be16(b, s + 3),
                         comps=[[b[s + 6 + 3 * c], b[s + 7 + 3 * c] >> 4, b[s + 7 + 3 * c] & 15, b[s + 8 + 3 * c], 0, 0] for c in range(nf)])
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported("progressive / lossless / arithmetic-coded JPEG")
        elif m == 0xC4:
            while s < e:
                tc, th = b[s] >> 4, b[s] & 15
                counts = bytes(b[s + 1:s + 17])
                ns = sum(counts)
                (ac if tc else dc)[th] = (counts, bytes(b[s + 17:s + 17 + ns]))
                s += 17 + ns
        elif m == 0xDD:
            ri = _be16(b, s)
        elif m == 0xE0 and e - s >= 5 and bytes(b[s:s + 5]) == b"JFIF\0":
            saw_jfif = True
        elif m == 0xEE and e - s >= 12 and bytes(b[s:s + 5]) == b"Adobe":
            adobe_transform = b[s + 11]
        elif m == 0xDA:
            if frame is None:
                raise Unsupported("scan before frame header")
            ns = b[s]
            comps = frame["comps"]
            if ns != len(comps):
                raise Unsupported("multi-scan (non-interleaved) file")
            for c in range(ns):
                if b[s + 1 + 2 * c] != comps[c][0]:
                    raise Unsupported("scan component order differs from the frame's")
                comps[c][4], comps[c][5] = b[s + 2 + 2 * c] >> 4, b[s + 2 + 2 * c] & 15
            if b[s + 1 + 2 * ns] != 0 or b[s + 2 + 2 * ns] != 63 or b[s + 3 + 2 * ns] != 0:
                raise Unsupported("spectral selection / successive approximation in a sequential scan")
            frame.update(qt=qt, dc=dc, ac=ac, restart_interval=ri, scan_off=e)
            break
        pos = e
    comps = frame["comps"]
    if frame["width"] <= 0 or frame["height"] <= 0:
        raise Unsupported("empty image")
    if len(comps) == 3:
        if (comps[1][1], comps[1][2], comps[2][1], comps[2][2]) != (1, 1, 1, 1):
            raise Unsupported("chroma sampling factors other than 1x1")
        if (comps[0][1], comps[0][2]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("luma sampling %dx%d" % (comps[0][1], comps[0][2]))
        # colour space as the IJG library decides it (jdapimin.c default_decompress_parms): JFIF -> YCbCr; else an Adobe marker's
        # transform flag (0 = the three components are RGB as stored); else component ids 'R','G','B' mean RGB
        if not saw_jfif:
            if (adobe_transform is not None and adobe_transform != 1) or \
                    (adobe_transform is None and [c[0] for c in comps] == [0x52, 0x47, 0x42]):
                raise Unsupported("RGB-coded JPEG")
    elif len(comps) == 1:
        comps[0][1] = comps[0][2] = 1            # a single-component scan is never interleaved
    else:
        raise Unsupported("%d components" % len(comps))
    for c in comps:
        if c[3] not in qt or c[4] not in dc or c[5] not in ac:
            raise Unsupported("missing quantisation / Huffman table")
    return frame


_HUFF_DT = np.dtype([("look", "<u2", 512), ("maxcode", "<i4", 18), ("valoff", "<i4", 17), ("sym", "u1", 256), ("pad", "u1", 4)])
assert _HUFF_DT.itemsize == HUFF_BYTES
_huff_cache = {}


def build_huff(counts, symbols):
    """(16 code counts, symbols) -> one spn_jpeg_huff record (bytes); cached: most files carry the standard tables."""
    key = (counts, symbols)
    rec = _huff_cache.get(key)
    if rec is not None:
        return rec
    t = np.zeros((), dtype=_HUFF_DT)
    t["maxcode"][:] = -1
    t["maxcode"][17] = 0x7FFFFFFF
    sym = np.frombuffer(symbols, dtype=np.uint8)
    if len(sym) > 256 or sum(counts) != len(sym):
        raise Unsupported("malformed Huffman table")
    t["sym"][:len(sym)] = sym
    code = k = 0
    for l in range(1, 17):
        c = counts[l - 1]
        if c:
            if code + c > (1 << l):
                raise Unsupported("over-subscribed Huffman table")
            t["valoff"][l] = k - code
            t["maxcode"][l] = code + c - 1
            if l <= 9:
                rep = 1 << (9 - l)
                entries = (np.uint16(l << 8) | sym[k:k + c].astype(np.uint16))
                t["look"][code << (9 - l):(code + c) << (9 - l)] = np.repeat(entries, rep)
            code += c
            k += c
        code <<= 1
    rec = t.tobytes()
    if len(_huff_cache) < 4096:
        _huff_cache[key] = rec
    return rec


class Batch:
    """Host-side description of a batch of decodable files: numpy arrays in the layouts of include/spn4cir_hip.h."""

    def __init__(self, files):
        frames = [parse_header(f) for f in files]
        n = len(files)
        self.n = n
        images = np.zeros((n, IMAGE_WORDS), dtype=np.uint32)
        segs, chunks, huffs, huff_index, qts, qt_index = [], [], [], {}, [], {}
        byte_off = coef_off = plane_off = rgb_off = 0
        self.sizes, self.max_blocks, self.max_pixels = [], 1, 1
        for i, (f, fr) in enumerate(zip(files, frames)):
            W, H, comps = fr["width"], fr["height"], fr["comps"]
            hs, vs = comps[0][1], comps[0][2]
            mcux, mcuy = -(-W // (8 * hs)), -(-H // (8 * vs))
            scan = bytes(f[fr["scan_off"]:])
            r = images[i]
            r[0:8] = (W, H, len(comps), hs, vs, mcux, mcuy, fr["restart_interval"])
            r[8:10] = (byte_off, len(scan))
            for c, comp in enumerate(comps):
                bx, by = mcux * (hs if c == 0 else 1), mcuy * (vs if c == 0 else 1)
                r[10 + c], r[13 + c], r[16 + c], r[19 + c] = coef_off, bx, by, plane_off
                coef_off += bx * by * 64
                plane_off += bx * by * 64
                self.max_blocks = max(self.max_blocks, bx * by)
                q = fr["qt"][comp[3]]
                qk = q.tobytes()
                if qk not in qt_index:
                    qt_index[qk] = len(qts)
                    qts.append(q)
                r[22 + c] = qt_index[qk]
                for slot, tabs, tid in ((25, fr["dc"], comp[4]), (28, fr["ac"], comp[5])):
                    rec = build_huff(*tabs[tid])
                    if rec not in huff_index:
                        huff_index[rec] = len(huffs)
                        huffs.append(rec)
                    r[slot + c] = huff_index[rec]
            r[31] = rgb_off
            rgb_off += W * H * 3
            rgb_off = (rgb_off + 15) & ~15
            self.sizes.append((H, W))
            self.max_pixels = max(self.max_pixels, W * H)
            # entropy segments: the whole scan, or one per restart interval (byte positions of the RSTn markers)
            nmcu, ri = mcux * mcuy, fr["restart_interval"]
            r[32] = len(segs)
            if ri and nmcu > ri:
                a = np.frombuffer(scan, dtype=np.uint8)
                mk = np.nonzero((a[:-1] == 0xFF) & (a[1:] >= 0xD0) & (a[1:] <= 0xD7))[0]
                want = -(-nmcu // ri) - 1
                if len(mk) < want:
                    raise Unsupported("fewer restart markers than restart intervals")
                starts = [0] + [int(p) + 2 for p in mk[:want]]
                for k, st in enumerate(starts):
                    segs.append((i, k * ri, min(ri, nmcu - k * ri), byte_off + st))
            else:
                segs.append((i, 0, nmcu, byte_off))
            r[33] = len(segs) - int(r[32])
            chunks.append(scan)
            byte_off += len(scan)
            if coef_off >= 1 << 31 or plane_off >= 1 << 31 or rgb_off >= 1 << 31 or byte_off >= 1 << 31:
                raise ValueError("batch too large for 32-bit offsets: decode fewer files per call")
        self.images = images
        self.segs = np.array(segs, dtype=np.uint32).reshape(-1, 4)
        self.bytes = np.frombuffer(bytearray(b"".join(chunks) + b"\xff\xd9\0\0"), dtype=np.uint8)
        self.huff = np.frombuffer(bytearray(b"".join(huffs)), dtype=np.uint8)
        self.qt = np.stack(qts).astype(np.uint16)
        self.coef_elems = (coef_off + 7) & ~7
        self.plane_bytes = plane_off
        self.rgb_bytes = max(rgb_off, 16)
        self.rgb_off = [int(x) for x in images[:, 31]]


def _pil_rgb(data):
    import io
    from PIL import Image
    return np.array(Image.open(io.BytesIO(bytes(data))).convert("RGB"), dtype=np.uint8)


RGB_BUDGET = 1 << 30        # bytes of decoded RGB per kernel batch (coefficients + planes add up to ~3x that on the device)


def split_by_pixels(sizes, budget=None):
    """[(width, height)] -> [[positions]]: consecutive runs whose decoded RGB stays under `budget` bytes (a single larger image
    gets a run of its own).  Keeps a chunk of a few thousand web photos (CIRR / NLVR2: several megapixels each) inside the
    kernels' 32-bit offsets and a bounded device footprint; the reference's per-image PIL path has no such limit
    (data_utils_negplus.py:298-304)."""
    budget = RGB_BUDGET if budget is None else budget
    runs, cur, acc = [], [], 0
    for k, (w, h) in enumerate(sizes):
        nb = 3 * int(w) * int(h)
        if cur and acc + nb > budget:
            runs.append(cur)
            cur, acc = [], 0
        cur.append(k)
        acc += nb
    if cur:
        runs.append(cur)
    return runs


def _decode_group(good, idx, out, fallback, device):
    """One kernel batch: `good` files at output positions `idx`; files a per-batch check rejects move to `fallback`."""
    from . import ops
    from ._lib import check, lib
    good, idx = list(good), list(idx)
    b = None
    while good:
        try:
            b = Batch(good)
            break
        except Unsupported:                      # a per-batch check failed (restart markers): find the file, decode it on the host
            for k in range(len(good)):
                try:
                    Batch([good[k]])
                except Unsupported:
                    fallback.append(idx[k])
                    del good[k], idx[k]
                    break
            else:
                raise
        except ValueError:                       # 32-bit offsets exhausted although the pixel budget held (odd sampling): halve
            if len(good) == 1:
                fallback.append(idx[0])
                return
            h = len(good) // 2
            _decode_group(good[:h], idx[:h], out, fallback, device)
            _decode_group(good[h:], idx[h:], out, fallback, device)
            return
    if not good:
        return
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    d_bytes, d_img, d_seg, d_huff, d_qt = up(b.bytes), up(b.images.view(np.int32)), up(b.segs.view(np.int32)), up(b.huff), \
        up(b.qt.view(np.int16))
    coefs = torch.empty(b.coef_elems, dtype=torch.int16, device=device)
    planes = torch.empty(max(b.plane_bytes, 16), dtype=torch.uint8, device=device)
    rgb = torch.empty(b.rgb_bytes, dtype=torch.uint8, device=device)
    check(lib().spn_jpeg_decode_batch(ops._p(d_bytes), ops._p(d_img), b.n, ops._p(d_seg), b.segs.shape[0], ops._p(d_huff),
                                      ops._p(d_qt), ops._p(coefs), b.coef_elems, ops._p(planes), ops._p(rgb), b.max_blocks,
                                      b.max_pixels, ops._stream()), "jpeg_decode_batch")
    for k, i in enumerate(idx):
        H, W = b.sizes[k]
        out[i] = rgb[b.rgb_off[k]:b.rgb_off[k] + H * W * 3].view(H, W, 3)


def decode_batch(files, device="cuda", rgb_budget=None):
    """files: list of bytes-like JPEG files -> list of uint8 [H, W, 3] tensors on `device` (views of a few batch buffers),
    equal to `np.asarray(PIL.Image.open(f).convert("RGB"))`.  Files outside the kernels' scope are decoded by Pillow on the host
    and uploaded (the second return value lists their positions).  The list is cut into kernel batches of at most `rgb_budget`
    bytes of decoded RGB (split_by_pixels), so any number of files of any size may be passed."""
    device = torch.device(device)
    out, fallback, good, idx, sizes = [None] * len(files), [], [], [], []
    for i, f in enumerate(files):
        try:
            hd = parse_header(f)
            good.append(f)
            idx.append(i)
            sizes.append((hd["width"], hd["height"]))
        except Unsupported:
            fallback.append(i)
    for run in split_by_pixels(sizes, rgb_budget):
        _decode_group([good[k] for k in run], [idx[k] for k in run], out, fallback, device)
    for i in fallback:
        out[i] = torch.from_numpy(_pil_rgb(files[i])).to(device)
    return out, sorted(fallback)
