// Baseline JPEG decode on the GPU: file bytes -> RGB uint8 [H][W][3] in HBM, for batches of images (the bank builders and
// extract_index_features decode every gallery image once: clip4cir/models_negplus.py:59-125, utils.py:24-50; the reference decodes on
// the host with PIL inside DataLoader workers, data_utils_negplus.py:268-319).  Per-work-item arithmetic: jpeg_core.h (also compiled
// for the host by the unit-test harness and checked against Pillow bit for bit).
//
// Three launches per batch behind one zero fill of the coefficient buffer:
//   jpeg_huffman_kernel   one LANE per entropy segment (a whole scan, or one restart interval): Huffman decoding is serial inside a
//                         segment, so the parallelism is across the images of the batch - 64 segments per wave, bytes from L2;
//                         non-zero coefficients are written in natural order as int16
//   jpeg_idct_kernel      one thread per 8 x 8 block: dequantise + islow IDCT -> uint8 planes (padded MCU grid)
//   jpeg_color_kernel     one thread per output pixel: fancy chroma upsampling + YCbCr -> RGB
// HBM-bound integer work: nothing here is reshaped into a GEMM.
#include "common.h"
#include "jpeg_core.h"
#include "kernels.h"

#define SPN_TRYJ(x)           \
    do {                      \
        int rc_ = (x);        \
        if (rc_) return rc_;  \
    } while (0)

namespace spn {

using spnjpeg::Huff;
using spnjpeg::Image;
using spnjpeg::Segment;

// Lanes per workgroup of the entropy kernel.  A lane's byte stream, table look-ups and block stores are its own, so a 64-lane
// wave instruction touches up to 64 cache lines; with 16 lanes per workgroup a batch of 256 files spreads over 16 CUs' memory pipes
// (each lane runs the same serial loop either way: lanes are the parallelism, not waves).
static constexpr int JPEG_LANES = 16;
static constexpr int JPEG_LDS = JPEG_LANES * 6 * (int)sizeof(Huff);
static_assert(sizeof(Huff) % 16 == 0 && JPEG_LDS <= 160 * 1024, "entropy kernel LDS");

__global__ __launch_bounds__(JPEG_LANES) void jpeg_huffman_kernel(const uint8_t* __restrict__ bytes, const Image* __restrict__ images,
                                                         const Segment* __restrict__ segs, int n_segs, const Huff* __restrict__ tabs,
                                                         int16_t* __restrict__ coefs) {
    // LDS of a workgroup: every lane's six Huffman tables (134 KB for 16 lanes) - nothing on a lane's serial path reads global
    // memory except its own byte stream (one prefetched dword per 32 bits); coefficients leave as 2-byte stores nobody waits for
    extern __shared__ __attribute__((aligned(16))) char jsm[];
    Huff* stab = (Huff*)jsm;                                                    // [JPEG_LANES][6]
    const int lane = threadIdx.x;
    const int s = blockIdx.x * JPEG_LANES + lane;
    const bool live = s < n_segs;
    Segment sg = {0, 0, 0, 0};
    Image im;
    if (live) {
        sg = segs[s];
        im = images[sg.image];
        for (int c = 0; c < im.ncomp; ++c) {
            const uint4* src_d = (const uint4*)(tabs + im.dc_tab[c]);
            const uint4* src_a = (const uint4*)(tabs + im.ac_tab[c]);
            uint4* dst_d = (uint4*)(stab + lane * 6 + 2 * c);
            uint4* dst_a = (uint4*)(stab + lane * 6 + 2 * c + 1);
            for (int i = 0; i < (int)(sizeof(Huff) / 16); ++i) { dst_d[i] = src_d[i]; dst_a[i] = src_a[i]; }
        }
    }
    __syncthreads();
    if (!live) return;
    spnjpeg::decode_segment(im, sg, bytes, stab + lane * 6, coefs);
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const Image* __restrict__ images, const int16_t* __restrict__ coefs,
                                                       const uint16_t* __restrict__ qtabs, uint8_t* __restrict__ planes) {
    const int ic = blockIdx.y, img = ic / 3, c = ic - img * 3;
    const Image& im = images[img];
    if (c >= im.ncomp) return;
    const int nb = im.blocks_x[c] * im.blocks_y[c];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const int by = b / im.blocks_x[c], bx = b - by * im.blocks_x[c];
    const int stride = im.blocks_x[c] * 8;
    spnjpeg::idct_block(coefs + im.coef_off[c] + (size_t)b * 64, qtabs + im.qt[c] * 64,
                        planes + im.plane_off[c] + (size_t)by * 8 * stride + bx * 8, stride);
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const Image* __restrict__ images, const uint8_t* __restrict__ planes,
                                                        uint8_t* __restrict__ rgb) {
    const Image& im = images[blockIdx.y];
    const int n = im.width * im.height;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i / im.width, x = i - y * im.width;
        uint8_t px[3];
        spnjpeg::pixel_rgb(im, planes, x, y, px);
        uint8_t* o = rgb + im.rgb_off + (size_t)i * 3;
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
    }
}

int jpeg_decode_batch(const uint8_t* bytes, const void* images, int n_images, const void* segs, int n_segs, const void* huff,
                      const uint16_t* qtabs, int16_t* coefs, size_t coef_elems, uint8_t* planes, uint8_t* rgb, int max_blocks,
                      int max_pixels, hipStream_t st) {
    if (!bytes || !images || !segs || !huff || !qtabs || !coefs || !planes || !rgb) return SPN_ERR_ARG;
    if (n_images <= 0 || n_segs <= 0 || max_blocks <= 0 || max_pixels <= 0) return SPN_ERR_ARG;
    if (coef_elems % 2 || ((uintptr_t)coefs & 15)) return SPN_ERR_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)jpeg_huffman_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, JPEG_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    SPN_TRYJ(zero_fill_f32((float*)coefs, coef_elems / 2, st));          // int16 pairs as 32-bit words: only non-zero coefficients are stored
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((n_segs + JPEG_LANES - 1) / JPEG_LANES), dim3(JPEG_LANES), JPEG_LDS, st, bytes, (const Image*)images, (const Segment*)segs,
                       n_segs, (const Huff*)huff, coefs);
    SPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((max_blocks + 255) / 256, n_images * 3), dim3(256), 0, st, (const Image*)images, coefs, qtabs,
                       planes);
    SPN_CHECK_LAUNCH();
    int gx = (max_pixels + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(gx, n_images), dim3(256), 0, st, (const Image*)images, planes, rgb);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
