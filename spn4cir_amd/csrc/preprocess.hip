// Image preprocessing of the bank builders / validation: the reference's
// targetpad_transform (clip4cir/data_utils.py:42-65,84-98) = TargetPad -> Resize(dim, BICUBIC) -> CenterCrop(dim)
// -> ToTensor -> Normalize, which torchvision runs through Pillow.  Integer work, bit-exact by construction:
// Pillow's two-pass 8-bit resampler (ImagingResample: per output pixel a window [xmin, xmin+n) of fixed-point
// coefficients with 22 fractional bits, sum seeded with 1 << 21, clip8(sum >> 22); horizontal pass first, its
// uint8 result feeds the vertical pass).  The coefficient tables come from the host (spn4cir_amd/preprocess.py,
// double arithmetic as in Pillow's precompute_coeffs); the zero padding of TargetPad is virtual (reads outside
// the source image return 0) and only the centre-cropped window is ever computed.
#include "common.h"
#include "kernels.h"

namespace spn {

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// tmp[y][x][c], y in [0, Hp) rows of the PADDED image, x in [0, dim) cropped output columns
__global__ void resize_h_kernel(const uint8_t* __restrict__ src, int H, int W, int hp, int vp, int Hp,
                                const int32_t* __restrict__ kx, const int32_t* __restrict__ bx, int ksize, int crop_left,
                                int dim, uint8_t* __restrict__ tmp) {
    const int total = Hp * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / dim, x = i % dim;
        const int sy = y - vp;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        if (sy >= 0 && sy < H) {
            const int xo = x + crop_left;
            const int x0 = bx[2 * xo], n = bx[2 * xo + 1];
            const int32_t* k = kx + (size_t)xo * ksize;
            const uint8_t* row = src + (size_t)sy * W * 3;
            for (int t = 0; t < n; ++t) {
                const int sx = x0 + t - hp;
                if (sx >= 0 && sx < W) {
                    const int w = k[t];
                    s0 += row[sx * 3 + 0] * w;
                    s1 += row[sx * 3 + 1] * w;
                    s2 += row[sx * 3 + 2] * w;
                }
            }
        }
        uint8_t* o = tmp + (size_t)i * 3;
        o[0] = (uint8_t)clip8(s0 >> 22);
        o[1] = (uint8_t)clip8(s1 >> 22);
        o[2] = (uint8_t)clip8(s2 >> 22);
    }
}

// out[c][y][x] = (clip8(vertical pass) / 255 - mean[c]) / std[c]   (ToTensor + Normalize, fp32, IEEE division)
__global__ void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int Hp, const int32_t* __restrict__ ky,
                                     const int32_t* __restrict__ by, int ksize, int crop_top, int dim, float m0, float m1,
                                     float m2, float sd0, float sd1, float sd2, float* __restrict__ out,
                                     uint8_t* __restrict__ out_u8) {
    const int total = dim * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / dim, x = i % dim;
        const int yo = y + crop_top;
        const int y0 = by[2 * yo], n = by[2 * yo + 1];
        const int32_t* k = ky + (size_t)yo * ksize;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int t = 0; t < n; ++t) {
            const uint8_t* p = tmp + ((size_t)(y0 + t) * dim + x) * 3;
            const int w = k[t];
            s0 += p[0] * w;
            s1 += p[1] * w;
            s2 += p[2] * w;
        }
        const int v0 = clip8(s0 >> 22), v1 = clip8(s1 >> 22), v2 = clip8(s2 >> 22);
        if (out_u8) {
            out_u8[(size_t)i * 3 + 0] = (uint8_t)v0;
            out_u8[(size_t)i * 3 + 1] = (uint8_t)v1;
            out_u8[(size_t)i * 3 + 2] = (uint8_t)v2;
        }
        if (out) {
            out[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.0f), m0), sd0);
            out[total + i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.0f), m1), sd1);
            out[2 * total + i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.0f), m2), sd2);
        }
    }
}

int preprocess_image(const uint8_t* src, int H, int W, int hp, int vp, const int32_t* kx, const int32_t* bx, int ksize_x,
                     const int32_t* ky, const int32_t* by, int ksize_y, int crop_left, int crop_top, int dim,
                     const float* mean3, const float* std3, uint8_t* tmp, float* out, uint8_t* out_u8, hipStream_t st) {
    if (!src || !kx || !bx || !ky || !by || !tmp || (!out && !out_u8) || !mean3 || !std3) return SPN_ERR_ARG;
    if (H <= 0 || W <= 0 || hp < 0 || vp < 0 || dim <= 0 || ksize_x <= 0 || ksize_y <= 0) return SPN_ERR_ARG;
    const int Hp = H + 2 * vp;
    const int n1 = Hp * dim, n2 = dim * dim;
    hipLaunchKernelGGL(resize_h_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, src, H, W, hp, vp, Hp, kx, bx, ksize_x,
                       crop_left, dim, tmp);
    SPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((n2 + 255) / 256), dim3(256), 0, st, tmp, Hp, ky, by, ksize_y, crop_top, dim,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out, out_u8);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
