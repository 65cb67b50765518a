"""Oracle: baseline JPEG -> RGB uint8, bit for bit as Pillow / libjpeg-turbo decode it (TEST INFRASTRUCTURE, like everything
under oracle/).

The reference decodes every image on the host with `PIL.Image.open(path)` + `.convert("RGB")` inside its DataLoader
workers (clip4cir/data_utils_negplus.py:17,268-319) before `targetpad_transform`.  PIL's decoder is libjpeg-turbo (third
party, not under /root/reference; Pillow 12.2 / libjpeg-turbo API level 6.2 in this image); its published algorithms are
restated here from ITU-T T.81 and the IJG library's documented decompression pipeline at its DEFAULT settings (which Pillow
uses): Huffman entropy decoding, dequantisation, the "islow" integer inverse DCT (Loeffler-Ligtenberg-Moshovitz, 13-bit
constants), "fancy" (triangle-filter) chroma upsampling for 2:1 horizontal and 2:1 x 2:1 subsampling with edge replication,
and the fixed-point YCbCr -> RGB conversion.  Pinned against PIL itself on a corpus of generated files
(tests/test_jpeg_cpu.py: sizes that are not MCU multiples, 4:4:4 / 4:2:2 / 4:2:0 / grayscale, qualities 30-100, restart
intervals, optimised Huffman tables).

Supported: baseline / extended-sequential Huffman (SOF0 / SOF1), 8-bit, 1 or 3 components in ONE interleaved scan, sampling
factors 1x1 for chroma and 1x1 / 2x1 / 2x2 for luma.  Everything else (progressive, arithmetic coding, CMYK, 4:4:0, 4:1:1,
multi-scan) raises Unsupported - the product falls back to PIL on the host for those files.
"""
import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                   54, 47, 55, 62, 63], dtype=np.int32)


class Unsupported(ValueError):
    pass


def parse(data):
    """-> dict(width, height, comps=[dict(id, h, v, tq, td, ta)], qt={tq: int32[64] natural order}, dc / ac = {th: (counts[16],
    symbols)}, restart_interval, scan = offset of the entropy-coded segment).  Raises Unsupported for what the kernels do not do."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG (no SOI)")
    pos = 2
    info = dict(qt={}, dc={}, ac={}, restart_interval=0, comps=None)
    n = len(data)
    while pos + 4 <= n:
        if data[pos] != 0xFF:
            raise Unsupported("marker expected")
        while pos < n and data[pos] == 0xFF:            # fill bytes
            pos += 1
        m = data[pos]
        pos += 1
        if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            break
        L = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + L]
        if m == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    vals = [(seg[i + 2 * k] << 8) | seg[i + 2 * k + 1] for k in range(64)]
                    i += 128
                else:
                    vals = list(seg[i:i + 64])
                    i += 64
                q = np.zeros(64, dtype=np.int32)
                q[ZIGZAG] = vals
                info["qt"][tq] = q
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise Unsupported("sample precision != 8")
            info["height"], info["width"] = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            nf = seg[5]
            info["comps"] = [dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c]) for c in range(nf)]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("progressive / lossless / arithmetic JPEG")
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                ns = sum(counts)
                info["ac" if tc else "dc"][th] = (counts, list(seg[i + 17:i + 17 + ns]))
                i += 17 + ns
        elif m == 0xDD:
            info["restart_interval"] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            if info["comps"] is None:
                raise Unsupported("SOS before SOF")
            ns = seg[0]
            if ns != len(info["comps"]):
                raise Unsupported("multi-scan (non-interleaved) file")
            for c in range(ns):
                cid, t = seg[1 + 2 * c], seg[2 + 2 * c]
                comp = next((x for x in info["comps"] if x["id"] == cid), None)
                if comp is None:
                    raise Unsupported("scan component not in frame")
                comp["td"], comp["ta"] = t >> 4, t & 15
            if [x["id"] for x in info["comps"]] != [seg[1 + 2 * c] for c in range(ns)]:
                raise Unsupported("scan component order differs from the frame's")
            info["scan"] = pos + L
            break
        pos += L
    if "scan" not in info:
        raise Unsupported("no scan")
    comps = info["comps"]
    if len(comps) not in (1, 3):
        raise Unsupported(f"{len(comps)} components")
    if len(comps) == 3:
        if (comps[1]["h"], comps[1]["v"], comps[2]["h"], comps[2]["v"]) != (1, 1, 1, 1):
            raise Unsupported("chroma sampling factors other than 1x1")
        if (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("luma sampling %dx%d" % (comps[0]["h"], comps[0]["v"]))
    else:
        comps[0]["h"] = comps[0]["v"] = 1               # a single-component scan is never interleaved: sampling factors are moot
    for c in comps:
        if c["tq"] not in info["qt"] or c["td"] not in info["dc"] or c["ta"] not in info["ac"]:
            raise Unsupported("missing table")
    return info


def _huff_lookup(counts, symbols):
    """code -> (length, symbol) as mincode / maxcode / valptr per length (T.81 F.2.2.3)."""
    mincode, maxcode, valptr = [0] * 17, [-1] * 17, [0] * 17
    code = k = 0
    for l in range(1, 17):
        valptr[l] = k
        mincode[l] = code
        code += counts[l - 1]
        k += counts[l - 1]
        maxcode[l] = code - 1 if counts[l - 1] else -1
        code <<= 1
    return mincode, maxcode, valptr, symbols


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        d = self.d
        if self.p < len(d):
            b = d[self.p]
            if b == 0xFF:
                nxt = d[self.p + 1] if self.p + 1 < len(d) else 0xD9
                if nxt == 0:
                    self.p += 2
                else:                                   # a marker: the decoder feeds zero bits (libjpeg's "insufficient data")
                    b = 0
            else:
                self.p += 1
        else:
            b = 0
        self.acc = ((self.acc << 8) | b) & 0xFFFFFFFF
        self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        while self.n < k:
            self._fill()
        v = (self.acc >> (self.n - k)) & ((1 << k) - 1)
        self.n -= k
        return v

    def decode(self, tab):
        mincode, maxcode, valptr, symbols = tab
        code = 0
        for l in range(1, 17):
            code = (code << 1) | self.get(1)
            if maxcode[l] >= 0 and code <= maxcode[l] and code >= mincode[l]:
                return symbols[valptr[l] + code - mincode[l]]
        return 0                                        # corrupt code: libjpeg warns and returns 0

    def restart(self):
        """byte-align and step over the RSTn marker"""
        self.acc = self.n = 0
        d = self.d
        while self.p + 1 < len(d) and not (d[self.p] == 0xFF and 0xD0 <= d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _extend(v, s):
    return v - (1 << s) + 1 if s and v < (1 << (s - 1)) else v


def entropy_decode(data, info):
    """-> list per component of int16 [blocks_y, blocks_x, 64] dequantised-ORDER (natural order) raw coefficients (not yet
    dequantised), in the padded MCU grid."""
    comps = info["comps"]
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcux, mcuy = -(-info["width"] // (8 * hmax)), -(-info["height"] // (8 * vmax))
    out = [np.zeros((mcuy * c["v"], mcux * c["h"], 64), dtype=np.int16) for c in comps]
    dct = {k: _huff_lookup(*v) for k, v in info["dc"].items()}
    act = {k: _huff_lookup(*v) for k, v in info["ac"].items()}
    br = _Bits(data, info["scan"])
    pred = [0] * len(comps)
    ri = info["restart_interval"]
    count = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if ri and count and count % ri == 0:
                br.restart()
                pred = [0] * len(comps)
            count += 1
            for ci, c in enumerate(comps):
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = out[ci][my * c["v"] + by, mx * c["h"] + bx]
                        t = br.decode(dct[c["td"]])
                        pred[ci] += _extend(br.get(t), t)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = br.decode(act[c["ta"]])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            if k > 63:
                                break
                            blk[ZIGZAG[k]] = _extend(br.get(s), s)
                            k += 1
    return out


F = dict(c0298=2446, c0390=3196, c0541=4433, c0765=6270, c0899=7373, c1175=9633, c1501=12299, c1847=15137, c1961=16069, c2053=16819,
         c2562=20995, c3072=25172)


def _idct_1d(x, shift, pass2):
    """x int64 [..., 8] -> int64 [..., 8]: one pass of the islow IDCT (13-bit constants) with its descale."""
    x0, x1, x2, x3, x4, x5, x6, x7 = [x[..., i] for i in range(8)]
    z1 = (x2 + x6) * F["c0541"]
    tmp2 = z1 + x6 * (-F["c1847"])
    tmp3 = z1 + x2 * F["c0765"]
    tmp0 = (x0 + x4) << 13
    tmp1 = (x0 - x4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = x7, x5, x3, x1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F["c1175"]
    t0, t1, t2, t3 = t0 * F["c0298"], t1 * F["c2053"], t2 * F["c3072"], t3 * F["c1501"]
    z1, z2, z3, z4 = z1 * (-F["c0899"]), z2 * (-F["c2562"]), z3 * (-F["c1961"]) + z5, z4 * (-F["c0390"]) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    rnd = 1 << (shift - 1)
    outs = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([(o + rnd) >> shift for o in outs], axis=-1)


def idct_blocks(coef, q):
    """coef int16 [by, bx, 64] (natural order) x quantiser q int32 [64] -> uint8 plane [8 by, 8 bx]."""
    by, bx, _ = coef.shape
    x = (coef.astype(np.int64) * q.astype(np.int64)).reshape(by, bx, 8, 8)          # [.., row, col]
    ws = _idct_1d(x.transpose(0, 1, 3, 2), 13 - 2, False)                               # pass 1 runs down the COLUMNS: [.., col, row']
    ws = ws.transpose(0, 1, 3, 2)                                                      # [.., row', col]
    out = _idct_1d(ws, 13 + 2 + 3, True)                                               # pass 2 along the rows
    out = np.clip(out + 128, 0, 255).astype(np.uint8)                                  # the SIMD decoders saturate
    return out.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def upsample_h2v1(plane, dw):
    """fancy 2:1 horizontal upsampling of the first dw columns -> [rows, 2 dw]; plain replication when dw <= 2."""
    p = plane[:, :dw].astype(np.int32)
    if dw <= 2:
        return np.repeat(p, 2, axis=1).astype(np.uint8)
    out = np.empty((p.shape[0], 2 * dw), dtype=np.int32)
    out[:, 0] = p[:, 0]
    out[:, 1] = (p[:, 0] * 3 + p[:, 1] + 2) >> 2
    mid = p[:, 1:-1] * 3
    out[:, 2:-2:2] = (mid + p[:, :-2] + 1) >> 2
    out[:, 3:-2:2] = (mid + p[:, 2:] + 2) >> 2
    out[:, -2] = (p[:, -1] * 3 + p[:, -2] + 1) >> 2
    out[:, -1] = p[:, -1]
    return out.astype(np.uint8)


def upsample_h2v2(plane, dw, dh):
    """fancy 2:1 x 2:1 upsampling of the [dh, dw] chroma plane (rows above the first / below the last replicate) -> [2 dh, 2 dw]."""
    p = plane[:dh, :dw].astype(np.int32)
    if dw <= 2:
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1).astype(np.uint8)
    above = np.vstack([p[:1], p[:-1]])
    below = np.vstack([p[1:], p[-1:]])
    out = np.empty((2 * dh, 2 * dw), dtype=np.int32)
    for v, nb in ((0, above), (1, below)):
        cs = p * 3 + nb                                 # column sums
        row = np.empty((dh, 2 * dw), dtype=np.int32)
        row[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        row[:, 1] = (cs[:, 0] * 3 + cs[:, 1] + 7) >> 4
        row[:, 2:-2:2] = (cs[:, 1:-1] * 3 + cs[:, :-2] + 8) >> 4
        row[:, 3:-2:2] = (cs[:, 1:-1] * 3 + cs[:, 2:] + 7) >> 4
        row[:, -2] = (cs[:, -1] * 3 + cs[:, -2] + 8) >> 4
        row[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2] = row
    return out.astype(np.uint8)


def _fix(x):
    return int(x * 65536 + 0.5)


def ycc_to_rgb(y, cb, cr):
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((_fix(1.40200) * cr + 32768) >> 16)
    g = y + ((-_fix(0.34414) * cb + 32768 - _fix(0.71414) * cr) >> 16)
    b = y + ((_fix(1.77200) * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(data):
    """JPEG file bytes -> uint8 [H, W, 3], as PIL.Image.open(...).convert("RGB")."""
    data = bytes(data)
    info = parse(data)
    coefs = entropy_decode(data, info)
    comps, W, H = info["comps"], info["width"], info["height"]
    planes = [idct_blocks(c, info["qt"][comps[i]["tq"]]) for i, c in enumerate(coefs)]
    if len(comps) == 1:
        g = planes[0][:H, :W]
        return np.stack([g, g, g], axis=-1)
    h, v = comps[0]["h"], comps[0]["v"]
    dw, dh = -(-W // h), -(-H // v)                     # downsampled chroma size
    if (h, v) == (1, 1):
        cb, cr = planes[1], planes[2]
    elif (h, v) == (2, 1):
        cb, cr = upsample_h2v1(planes[1], dw), upsample_h2v1(planes[2], dw)
    else:
        cb, cr = upsample_h2v2(planes[1], dw, dh), upsample_h2v2(planes[2], dw, dh)
    return ycc_to_rgb(planes[0][:H, :W], cb[:H, :W], cr[:H, :W])
