// bf16 fast path of CLIP's ModifiedResNet image towers (clip4cir/clip/model.py:10-155; RN50x4 is the argparse default of
// train_negplus.py:192): NHWC bf16 activations with the channel count padded to a multiple of 64 (pad channels are zero and
// stay zero: zero weight rows, zero bias, relu(0) = 0), every convolution = [im2col +] the bf16 MFMA gemm_nt with the
// eval-mode BatchNorm folded into weight and bias, ReLU / residual add as one elementwise pass.  The fp32 path in exact.hip
// stays the parity (exact) mode; this one is for bank extraction and validation throughput.
// im2col column order is tap-major, channel-minor: col = (ky * 3 + kx) * Cp + c, so a row is nine contiguous Cp-long copies.
#include "common.h"
#include "kernels.h"

namespace spn {

static inline unsigned rn_grid(size_t n) { return (unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256); }

// x [B, H, W, Cp] bf16 -> out [B * Ho * Wo, 9 * Cp]; one thread per 16-byte chunk (8 channels)
__global__ void im2col3x3_nhwc_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int Cp,
                                           int stride, int Ho, int Wo) {
    const int c8 = Cp >> 3;
    const size_t total = (size_t)B * Ho * Wo * 9 * c8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c8);
        const int tap = (int)((i / c8) % 9);
        const size_t row = i / ((size_t)9 * c8);
        const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), b = (int)(row / ((size_t)Wo * Ho));
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const u32x4*)(x + (((size_t)b * H + iy) * W + ix) * Cp + ch * 8);
        *(u32x4*)(out + row * (size_t)(9 * Cp) + (size_t)tap * Cp + ch * 8) = v;
    }
}

// the stem's first convolution: image fp32 NCHW [B, 3, H, W] -> out bf16 [B * Ho * Wo, 64], col = (ky * 3 + kx) * 3 + c
__global__ void im2col3x3_stem_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W, int stride,
                                           int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i & 63);
        const size_t row = i >> 6;
        float v = 0.f;
        if (col < 27) {
            const int tap = col / 3, c = col % 3;
            const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), b = (int)(row / ((size_t)Wo * Ho));
            const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * 3 + c) * H + iy) * W + ix];
        }
        out[i] = f2bf(v);
    }
}

// y = relu(y + r) (r optional), 8 elements per thread
__global__ void relu_add_bf16_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ r, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        bf16x8 a = *(const bf16x8*)(y + i * 8);
        if (r) {
            const bf16x8 b = *(const bf16x8*)(r + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = f2bf(fmaxf(bf2f(a[e]) + bf2f(b[e]), 0.f));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = f2bf(fmaxf(bf2f(a[e]), 0.f));
        }
        *(bf16x8*)(y + i * 8) = a;
    }
}

// nn.AvgPool2d(k) (kernel = stride = k) on NHWC bf16, fp32 accumulation
__global__ void avgpool_nhwc_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int Cp, int k) {
    const int Ho = H / k, Wo = W / k, c8 = Cp >> 3;
    const size_t total = (size_t)B * Ho * Wo * c8;
    const float inv = 1.0f / (float)(k * k);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c8);
        const size_t p = i / c8;
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const bf16x8 v = *(const bf16x8*)(x + (((size_t)b * H + oy * k + dy) * W + ox * k + dx) * Cp + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += bf2f(v[e]);
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(s[e] * inv);
        *(bf16x8*)(y + p * Cp + ch * 8) = o;
    }
}

int im2col3x3_nhwc_bf16(const bf16_t* x, bf16_t* out, int B, int H, int W, int Cp, int stride, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || Cp <= 0 || stride <= 0 || !x || !out) return SPN_ERR_ARG;
    if (Cp % 8) return SPN_ERR_SHAPE;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t n = (size_t)B * Ho * Wo * 9 * (Cp / 8);
    hipLaunchKernelGGL(im2col3x3_nhwc_bf16_kernel, dim3(rn_grid(n)), dim3(256), 0, st, x, out, B, H, W, Cp, stride, Ho, Wo);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int im2col3x3_stem_bf16(const float* image, bf16_t* out, int B, int H, int W, int stride, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || stride <= 0 || !image || !out) return SPN_ERR_ARG;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t n = (size_t)B * Ho * Wo * 64;
    hipLaunchKernelGGL(im2col3x3_stem_bf16_kernel, dim3(rn_grid(n)), dim3(256), 0, st, image, out, B, H, W, stride, Ho, Wo);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int relu_add_bf16(bf16_t* y, const bf16_t* resid, size_t n, hipStream_t st) {
    if (!y || n == 0) return SPN_ERR_ARG;
    if (n % 8) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(relu_add_bf16_kernel, dim3(rn_grid(n / 8)), dim3(256), 0, st, y, resid, n / 8);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int avgpool_nhwc_bf16(const bf16_t* x, bf16_t* y, int B, int H, int W, int Cp, int k, hipStream_t st) {
    if (B <= 0 || Cp <= 0 || k <= 0 || H < k || W < k || !x || !y) return SPN_ERR_ARG;
    if (Cp % 8) return SPN_ERR_SHAPE;
    const size_t n = (size_t)B * (H / k) * (W / k) * (Cp / 8);
    hipLaunchKernelGGL(avgpool_nhwc_bf16_kernel, dim3(rn_grid(n)), dim3(256), 0, st, x, y, B, H, W, Cp, k);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
