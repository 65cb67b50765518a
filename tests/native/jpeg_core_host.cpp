// Unit-test harness (TEST INFRASTRUCTURE): spn4cir_amd/csrc/jpeg_core.h - the per-work-item source of the GPU JPEG kernels - compiled
// for the host with g++ and driven by plain loops in the kernels' order, so that tests/test_jpeg_cpu.py can hold the exact arithmetic the
// device runs against Pillow without a GPU.  Never loaded by the product (spn4cir_amd has no CPU decode path).
#include <cstring>

#include "../../spn4cir_amd/csrc/jpeg_core.h"

using namespace spnjpeg;

extern "C" int jpeg_core_decode_host(const uint8_t* bytes, const Image* images, int n_images, const Segment* segs, int n_segs,
                                     const Huff* huff, const uint16_t* qtabs, int16_t* coefs, size_t coef_elems, uint8_t* planes,
                                     uint8_t* rgb) {
    std::memset(coefs, 0, coef_elems * sizeof(int16_t));             // the kernels' zero fill
    for (int s = 0; s < n_segs; ++s) {
        const Image& im = images[segs[s].image];
        Huff tabs6[6];                                                 // the kernel copies the same six tables into a lane's LDS slots
        for (int c = 0; c < im.ncomp; ++c) {
            tabs6[2 * c] = huff[im.dc_tab[c]];
            tabs6[2 * c + 1] = huff[im.ac_tab[c]];
        }
        decode_segment(im, segs[s], bytes, tabs6, coefs);
    }
    for (int i = 0; i < n_images; ++i) {
        const Image& im = images[i];
        for (int c = 0; c < im.ncomp; ++c) {
            const int stride = im.blocks_x[c] * 8;
            for (int b = 0; b < im.blocks_x[c] * im.blocks_y[c]; ++b) {
                const int by = b / im.blocks_x[c], bx = b - by * im.blocks_x[c];
                idct_block(coefs + im.coef_off[c] + (size_t)b * 64, qtabs + im.qt[c] * 64,
                           planes + im.plane_off[c] + (size_t)by * 8 * stride + bx * 8, stride);
            }
        }
        for (int y = 0; y < im.height; ++y)
            for (int x = 0; x < im.width; ++x) pixel_rgb(im, planes, x, y, rgb + im.rgb_off + ((size_t)y * im.width + x) * 3);
    }
    return 0;
}
