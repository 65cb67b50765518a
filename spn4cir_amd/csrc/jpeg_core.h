// Baseline JPEG decode, per-work-item cores shared by the device kernels (jpeg.hip) and the host unit-test harness
// (tests/native/jpeg_core_host.cpp compiles THIS file with g++ so that the exact source the GPU runs is checked against Pillow on
// the CPU; the harness is test infrastructure - the product has no CPU decode path).
//
// What is restated (Pillow = libjpeg-turbo at its default decompression settings, which is what the reference's
// `PIL.Image.open(path)` / `.convert("RGB")` runs inside its DataLoader workers, clip4cir/data_utils_negplus.py:17,268-319):
//   ITU-T T.81 Huffman entropy decoding (F.2.2), dequantisation, the IJG "islow" integer inverse DCT (13-bit constants, two
//   passes, descale 11 / 18 bits, +128, saturate), "fancy" triangle-filter chroma upsampling for h2v1 / h2v2 with edge
//   replication (plain replication when the chroma plane is <= 2 samples wide), fixed-point YCbCr -> RGB (16-bit scaled constants).
// CPU restatement of the same (numpy, independent code): oracle/jpeg_decode.py, pinned against Pillow.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SPN_JHD __host__ __device__ __forceinline__
#else
#define SPN_JHD inline
#endif

namespace spnjpeg {

// One image of a batch (mirrors spn_jpeg_image in include/spn4cir_hip.h; 40 x 32-bit words).
struct Image {
    int32_t width, height, ncomp, hs, vs;        // luma sampling factors (chroma is 1 x 1): 1x1, 2x1, 2x2
    int32_t mcux, mcuy, restart_interval;
    uint32_t scan_off, scan_len;                 // entropy-coded segment in the batch's byte buffer
    uint32_t coef_off[3];                        // int16 ELEMENT offset of each component's [blocks_y][blocks_x][64] coefficients (zigzag order)
    int32_t blocks_x[3], blocks_y[3];            // padded (whole-MCU) block grid per component
    uint32_t plane_off[3];                       // byte offset of each component's uint8 plane [8 blocks_y][8 blocks_x]
    int32_t qt[3], dc_tab[3], ac_tab[3];         // indices into the batch's quantiser / Huffman table arrays
    uint32_t rgb_off;                            // byte offset of the [height][width][3] output
    int32_t first_seg, n_seg;                    // this image's entropy segments in the segment array (restart intervals)
    int32_t reserved[6];
};
static_assert(sizeof(Image) == 40 * 4, "spn_jpeg_image layout");

// One independently decodable piece of an entropy-coded segment: a whole scan, or one restart interval.
struct Segment {
    int32_t image, mcu_first, mcu_count;
    uint32_t byte_off;                           // first byte of the piece (behind the RSTn marker)
};

// Huffman table in decode form: 9-bit look-ahead (entry = length << 8 | symbol, 0 = longer code), canonical slow path.
struct Huff {
    uint16_t look[512];
    int32_t maxcode[18];                         // [l] = largest code of length l (-1: none); [17] sentinel
    int32_t valoff[17];                          // [l] = index of the first symbol of length l minus its smallest code
    uint8_t sym[256];
    uint8_t pad[4];
};
static_assert(sizeof(Huff) == 1024 + 72 + 68 + 256 + 4, "spn_jpeg_huff layout");

SPN_JHD int zigzag(int k) {                      // k-th coefficient of the zigzag scan -> natural (row-major) position
    const unsigned char z[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                                 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46,
                                 53, 60, 61, 54, 47, 55, 62, 63};
    return z[k];
}

// 32-bit load from an arbitrary byte address, RAW (gfx950 global memory serves unaligned dwords: ONE global_load_dword), and its
// big-endian reading.  The two are separate on purpose: the byte swap at the point of USE lets the load issued at the end of one
// refill stay in flight until the next refill needs it (swapping where it is loaded puts the wait right behind the load).
SPN_JHD uint32_t load_raw32(const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *(const u32u*)p;
#else
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
#endif
}
SPN_JHD uint32_t be32(uint32_t raw) { return __builtin_bswap32(raw); }

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc;
    int n;
    uint32_t ahead;                              // the four bytes at p (raw), requested when p was set: latency hidden behind decoding
    SPN_JHD void init(const uint8_t* begin, const uint8_t* stop) {
        p = begin; end = stop; acc = 0; n = 0;
        ahead = (p + 4 <= end) ? load_raw32(p) : 0;
    }
    // Keep at least 33 valid bits.  Fast path: four data bytes at once when none of them is 0xFF (no stuffing, no marker).  Slow path,
    // byte by byte: FF 00 is a data byte 0xFF; any other FF xx is a marker - it ends the data: zero bits from there on, the
    // pointer stays on the marker (what the IJG decoder feeds after "premature end of data segment").
    SPN_JHD void fill() {
        const bool need = n <= 32;
        const uint32_t w = ahead;                                          // byte order does not matter for the 0xFF test
        const bool fast = need && (p + 4 <= end) && ((((~w) - 0x01010101u) & w & 0x80808080u) == 0);
        acc = fast ? ((acc << 32) | be32(w)) : acc;                        // selects, not a branch: see decode_segment
        n += fast ? 32 : 0;
        p += fast ? 4 : 0;
        if (need && !fast) {
            while (n <= 56) {
                unsigned b = 0;
                if (p < end) {
                    b = *p;
                    if (b == 0xFF) {
                        const unsigned nx = (p + 1 < end) ? p[1] : 0xD9u;
                        if (nx == 0) p += 2;
                        else b = 0;
                    } else {
                        ++p;
                    }
                }
                acc = (acc << 8) | b;
                n += 8;
            }
        }
        if (need) ahead = (p + 4 <= end) ? load_raw32(p) : 0;
    }
    SPN_JHD unsigned peek(int k) const { return (unsigned)((acc >> (n - k)) & ((1u << k) - 1u)); }
    SPN_JHD void skip(int k) { n -= k; }
};

// One Huffman symbol AND the value bits behind it from ONE 32-bit window of the stream (code <= 16 bits + value <= 15 bits): returns
// the symbol; *value = the sign-extended magnitude of (symbol & 15) bits (T.81 F.2.2.1 EXTEND), 0 when that is 0 bits.
// A lane's decode loop is a serial chain - on this GPU every dependent VALU instruction of a lone wave costs ~9-18 cycles and an LDS
// round trip ~64 (tools/lat/lat.hip) - so what counts is instructions per symbol: one refill test, one 64-bit shift, one look-up.
SPN_JHD int decode_symbol(BitReader& br, const Huff& t, int* value) {
    br.fill();                                                   // >= 33 valid bits
    const uint32_t w = (uint32_t)(br.acc >> (br.n - 32));        // the next 32 bits, left-aligned
    unsigned e = t.look[w >> 23];
    int len, sym;
    if (e) {
        len = (int)(e >> 8);
        sym = (int)(e & 255u);
    } else {
        len = 16;
        sym = 0;                                                 // not a code of this table (corrupt data): the IJG decoder substitutes 0
        for (int l = 10; l <= 16; ++l) {
            const int code = (int)(w >> (32 - l));
            if (code <= t.maxcode[l]) {
                len = l;
                sym = t.sym[(t.valoff[l] + code) & 255];
                break;
            }
        }
    }
    const int s = sym & 15;
    int v = (int)(((w << len) >> 1) >> (31 - s));                // s = 0: zero bits, v = 0 - no branch
    v += (v < ((1 << s) >> 1)) ? 1 - (1 << s) : 0;
    *value = v;
    br.n -= len + s;
    return sym;
}

// Entropy-decode one segment (a whole scan or one restart interval) as ONE flat loop over symbols.  tabs6 = the image's six Huffman
// tables in the order {dc, ac} of component 0, 1, 2 (the caller's fast copies - a lane's LDS slots on the device; unused slots of a
// grayscale image are never read).  Coefficients are stored in ZIGZAG order (the order of the stream: position k of a block = its
// k-th coded coefficient; the inverse DCT un-zigzags with compile-time indices), straight into the zero-filled coefficient buffer:
// one 2-byte store per non-zero coefficient, nothing to wait for.
// Why a flat loop: the lanes of a wave decode different files in lockstep.  With a loop per block the wave re-converges at every
// block end, so every block costs the LONGEST block among the lanes (measured: 1 357 cycles per symbol of lane 0 for 16 different
// photos side by side); here a lane's position (component, block, coefficient) is plain data, block changes are a few predicated
// integer instructions, and the wave runs max-over-lanes of the TOTAL symbol count.
SPN_JHD int decode_segment(const Image& im, const Segment& sg, const uint8_t* bytes, const Huff* tabs6, int16_t* coefs) {
    BitReader br;
    br.init(bytes + sg.byte_off, bytes + im.scan_off + im.scan_len);
    const int ncomp = im.ncomp, hs = im.hs, vs = im.vs, mcux = im.mcux;
    const int nb0 = hs * vs;                                   // luma blocks per MCU (chroma: one each)
    const int stride0 = im.blocks_x[0] * 64;                   // elements per block row of the luma plane
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int mcu_left = sg.mcu_count;
    const int my0 = sg.mcu_first / mcux;
    int mx = sg.mcu_first - my0 * mcux;
    // first block of the current MCU in each component's coefficient plane; stepping to the next MCU is an addition (a luma MCU
    // row is vs block rows: at the end of a row the pointer has advanced one block row by itself and skips vs - 1 more)
    int16_t* mb0 = coefs + im.coef_off[0] + ((size_t)(my0 * vs) * im.blocks_x[0] + mx * hs) * 64;
    int16_t* mb1 = coefs + im.coef_off[1] + ((size_t)my0 * im.blocks_x[1] + mx) * 64;
    int16_t* mb2 = coefs + im.coef_off[2] + ((size_t)my0 * im.blocks_x[2] + mx) * 64;
    const int row_skip0 = (vs - 1) * stride0;
    int c = 0, bi = 0, k = 0;                                  // component, block inside the MCU's component, coefficient index
    int16_t* blk = mb0;
    // The body is written with selects instead of nested branches on purpose: every `if` of a divergent loop costs the wave a
    // handful of exec-mask instructions whether or not a lane takes it, and instruction count is the price here (see decode_symbol).
    while (mcu_left > 0) {
        int v;
        const int sym = decode_symbol(br, tabs6[2 * c + (k != 0)], &v);
        const bool is_dc = k == 0;
        const int r = sym >> 4, s = sym & 15;
        const int pn = (c == 0 ? pred0 : (c == 1 ? pred1 : pred2)) + v;          // DC: the symbol is the category, v the difference
        pred0 = (is_dc && c == 0) ? pn : pred0;
        pred1 = (is_dc && c == 1) ? pn : pred1;
        pred2 = (is_dc && c == 2) ? pn : pred2;
        const int kk = is_dc ? 0 : k + r;                                           // AC: skip r zeros
        if (is_dc || (s != 0 && kk <= 63)) blk[kk] = (int16_t)(is_dc ? pn : v);
        const int kn = is_dc ? 1 : (s != 0 ? kk + 1 : (r == 15 ? k + 16 : 64));     // next index; ZRL : EOB
        const bool adv = kn >= 64;                                                  // block finished
        k = adv ? 0 : kn;
        const int bi2 = bi + (adv ? 1 : 0);
        const bool wb = bi2 >= (c == 0 ? nb0 : 1);                                  // component finished (only possible when adv)
        bi = wb ? 0 : bi2;
        const int cn = c + (wb ? 1 : 0);
        const bool wc = cn >= ncomp;                                                // MCU finished
        c = wc ? 0 : cn;
        const int mxn = mx + (wc ? 1 : 0);
        const bool wx = mxn == mcux;                                                // MCU row finished (only possible when wc)
        mx = wx ? 0 : mxn;
        mcu_left -= wc ? 1 : 0;
        mb0 += (wc ? hs * 64 : 0) + (wx ? row_skip0 : 0);
        mb1 += wc ? 64 : 0;
        mb2 += wc ? 64 : 0;
        const int by_off = (vs == 2 && (bi & 2)) ? stride0 : 0;                     // hs = vs = 2: block 2, 3 = second block row
        const int bx_off = (hs == 2 && (bi & 1)) ? 64 : 0;
        blk = c == 0 ? mb0 + by_off + bx_off : (c == 1 ? mb1 : mb2);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ inverse DCT
SPN_JHD void idct_1d(const int32_t* x, int stride, int32_t* o, int ostride, int shift) {
    const int32_t x0 = x[0], x1 = x[stride], x2 = x[2 * stride], x3 = x[3 * stride], x4 = x[4 * stride], x5 = x[5 * stride],
                  x6 = x[6 * stride], x7 = x[7 * stride];
    int32_t z1 = (x2 + x6) * 4433;
    const int32_t tmp2 = z1 + x6 * (-15137);
    const int32_t tmp3 = z1 + x2 * 6270;
    const int32_t tmp0 = (int32_t)((uint32_t)(x0 + x4) << 13);
    const int32_t tmp1 = (int32_t)((uint32_t)(x0 - x4) << 13);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int32_t t0 = x7, t1 = x5, t2 = x3, t3 = x1;
    z1 = t0 + t3;
    int32_t z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int32_t z5 = (z3 + z4) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1 *= -7373; z2 *= -20995;
    z3 = z3 * (-16069) + z5;
    z4 = z4 * (-3196) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int32_t rnd = 1 << (shift - 1);
    o[0] = (tmp10 + t3 + rnd) >> shift;
    o[7 * ostride] = (tmp10 - t3 + rnd) >> shift;
    o[ostride] = (tmp11 + t2 + rnd) >> shift;
    o[6 * ostride] = (tmp11 - t2 + rnd) >> shift;
    o[2 * ostride] = (tmp12 + t1 + rnd) >> shift;
    o[5 * ostride] = (tmp12 - t1 + rnd) >> shift;
    o[3 * ostride] = (tmp13 + t0 + rnd) >> shift;
    o[4 * ostride] = (tmp13 - t0 + rnd) >> shift;
}

// coef [64] x q [64], BOTH in zigzag order (as coded in the file) -> out [8 rows][8] uint8 at `stride` bytes per row
SPN_JHD void idct_block(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
    int32_t in[64], ws[64];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 64; ++i) in[zigzag(i)] = (int32_t)coef[i] * (int32_t)q[i];
    for (int c = 0; c < 8; ++c) idct_1d(in + c, 8, ws + c, 8, 13 - 2);               // pass 1: down each column
    for (int r = 0; r < 8; ++r) {
        int32_t row[8];
        idct_1d(ws + r * 8, 1, row, 1, 13 + 2 + 3);                                   // pass 2: along each row
        for (int c = 0; c < 8; ++c) {
            int v = row[c] + 128;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out[(size_t)r * stride + c] = (uint8_t)v;
        }
    }
}

// ------------------------------------------------------------------------------------ upsampling + colour conversion
SPN_JHD int chroma_at(const Image& im, const uint8_t* plane, int pstride, int x, int y) {
    if (im.hs == 1) return plane[(size_t)y * pstride + x];                               // 4:4:4
    const int dw = (im.width + 1) >> 1;                                                  // downsampled width (hs == 2)
    const int cx = x >> 1;
    if (im.vs == 1) {                                                                    // h2v1
        const uint8_t* row = plane + (size_t)y * pstride;
        const int p = row[cx];
        if (dw <= 2) return p;
        if (x == 0 || x == 2 * dw - 1) return p;
        return (x & 1) ? (3 * p + row[cx + 1] + 2) >> 2 : (3 * p + row[cx - 1] + 1) >> 2;
    }
    const int dh = (im.height + 1) >> 1;                                                 // h2v2
    const int cy = y >> 1;
    if (dw <= 2) return plane[(size_t)cy * pstride + cx];
    int ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : (ny > dh - 1 ? dh - 1 : ny);
    const uint8_t* r0 = plane + (size_t)cy * pstride;
    const uint8_t* r1 = plane + (size_t)ny * pstride;
    const int cs = 3 * r0[cx] + r1[cx];
    if (x == 0) return (cs * 4 + 8) >> 4;
    if (x == 2 * dw - 1) return (cs * 4 + 7) >> 4;
    if (x & 1) return (cs * 3 + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
    return (cs * 3 + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
}

SPN_JHD int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// pixel (x, y) of image im -> rgb[3]
SPN_JHD void pixel_rgb(const Image& im, const uint8_t* planes, int x, int y, uint8_t* rgb) {
    const int Y = planes[im.plane_off[0] + (size_t)y * (im.blocks_x[0] * 8) + x];
    if (im.ncomp == 1) {
        rgb[0] = rgb[1] = rgb[2] = (uint8_t)Y;
        return;
    }
    const int cb = chroma_at(im, planes + im.plane_off[1], im.blocks_x[1] * 8, x, y) - 128;
    const int cr = chroma_at(im, planes + im.plane_off[2], im.blocks_x[2] * 8, x, y) - 128;
    rgb[0] = (uint8_t)clamp255(Y + ((91881 * cr + 32768) >> 16));
    rgb[1] = (uint8_t)clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    rgb[2] = (uint8_t)clamp255(Y + ((116130 * cb + 32768) >> 16));
}

}  // namespace spnjpeg
