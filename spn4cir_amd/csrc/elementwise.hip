// HBM-bound helper kernels: casts, transposes, column sums, embedding gather/scatter,
// EOT pooling, fused AdamW.  All are streaming kernels: 16-byte accesses per lane,
// grid-stride loops, grids capped at 2048 blocks (cdna guide G11/G13).
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static inline int grid_for(size_t work_items, int block = 256, int cap = 2048) {
    size_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    return (int)(b > (size_t)cap ? cap : b);
}

// ------------------------------------------------------------------------------- casts
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    const size_t n8 = n >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(x + i * 8), b = *(const f32x4*)(x + i * 8 + 4);
        bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
        *(bf16x8*)(y + i * 8) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[(n8 << 3) + threadIdx.x] = f2bf(x[(n8 << 3) + threadIdx.x]);
}

__global__ void scale_cast_bf16_kernel(const float* __restrict__ x, const float* __restrict__ scale_dev, int reciprocal,
                                       bf16_t* __restrict__ y, int B, int D, int ldo) {
    const float s = reciprocal ? 1.0f / *scale_dev : *scale_dev;
    const size_t n = (size_t)B * ldo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / ldo), c = (int)(i % ldo);
        y[i] = c < D ? f2bf(x[(size_t)b * D + c] * s) : (bf16_t)0.f;
    }
}

int scale_cast_bf16(const float* x, const float* scale_dev, int reciprocal, bf16_t* y, int B, int D, int ldo, hipStream_t st) {
    if (B <= 0 || D <= 0 || ldo < D) return SPN_ERR_ARG;
    const size_t n = (size_t)B * ldo;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(scale_cast_bf16_kernel, dim3(blocks), dim3(256), 0, st, x, scale_dev, reciprocal, y, B, D, ldo);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// out[b, :] = bank[idx[b], :] for rows of `row_elems` bf16 (a multiple of 8): the reference-token gather of the BLIP step
// (blip4cir/models.py:97-100: refer_bank[refer_indexs] - there a host gather + a 227 MB upload per step) from a bf16 token bank
// that lives on the device, straight into the K/V projections' A operand.  An index outside the bank writes zeros (never
// dereferenced).  grid = (chunks per row, B): 16-byte loads / stores, every lane one piece per iteration.
__global__ __launch_bounds__(256) void gather_bank_rows_bf16_kernel(const bf16_t* __restrict__ bank, const int64_t* __restrict__ idx,
                                                                   int64_t n_rows, bf16_t* __restrict__ out, size_t row_elems) {
    const int b = blockIdx.y;
    const int64_t r = idx[b];
    const bool ok = r >= 0 && r < n_rows;
    const u32x4* src = (const u32x4*)(bank + (size_t)(ok ? r : 0) * row_elems);
    u32x4* dst = (u32x4*)(out + (size_t)b * row_elems);
    const size_t n16 = row_elems / 8;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = ok ? __builtin_nontemporal_load(src + i) : z;
}

int gather_bank_rows_bf16(const bf16_t* bank, const int64_t* idx, int64_t n_rows, bf16_t* out, int B, size_t row_elems,
                          hipStream_t st) {
    if (!bank || !idx || !out || B <= 0 || n_rows <= 0) return SPN_ERR_ARG;
    if (row_elems % 8 || ((uintptr_t)bank & 15) || ((uintptr_t)out & 15)) return SPN_ERR_SHAPE;
    const size_t n16 = row_elems / 8;
    int gx = (int)((n16 + 255) / 256);
    const int want = (2048 + B - 1) / B;                // ~8 workgroups per CU over the whole grid
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gather_bank_rows_bf16_kernel, dim3(gx, B), dim3(256), 0, st, bank, idx, n_rows, out, row_elems);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// Learnable temperature of the BLIP step (blip4cir/models.py:29, logits = q . bank / tau): with dqk = d loss / d (q / tau),
//   d loss / d tau = -(sum_b <q_b, dqk_b>) / tau^2   (times alpha and an optional device scalar: a caller whose gradient
//   is d loss / d q passes alpha = tau, autograd's incoming d(loss) as the scalar),   and 1 / tau for the kernels that scale by it.
// One workgroup, fixed summation order (thread = (row mod 4, column quad); then the 256-leaf tree): bit-reproducible.
__global__ __launch_bounds__(256) void tau_grad_kernel(const float* __restrict__ q, const float* __restrict__ dqk, int lddq,
                                                      const float* __restrict__ tau, int B, int D, float alpha,
                                                      const float* __restrict__ scale_dev, float* __restrict__ dtau,
                                                      float* __restrict__ inv_tau) {
    __shared__ float red[256];
    // thread = (row group tid / 64, column quad lane): 16-byte loads, four rows in flight per thread, no integer division
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    float acc = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        if (c + 4 <= D && (D % 4) == 0 && (lddq % 4) == 0) {
            int b = rg;
            for (; b + 12 < B; b += 16) {
                f32x4 a[4], g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[k] = *(const f32x4*)(q + (size_t)(b + 4 * k) * D + c);
                    g[k] = *(const f32x4*)(dqk + (size_t)(b + 4 * k) * lddq + c);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += a[k][0] * g[k][0] + a[k][1] * g[k][1] + a[k][2] * g[k][2] + a[k][3] * g[k][3];
            }
            for (; b < B; b += 4) {
                const f32x4 a = *(const f32x4*)(q + (size_t)b * D + c), g = *(const f32x4*)(dqk + (size_t)b * lddq + c);
                acc += a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
            }
        } else {
            for (int b = rg; b < B; b += 4)
                for (int e = c; e < min(c + 4, D); ++e) acc += q[(size_t)b * D + e] * dqk[(size_t)b * lddq + e];
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float it = 1.0f / tau[0];
        if (dtau) dtau[0] = -red[0] * it * it * alpha * (scale_dev ? scale_dev[0] : 1.0f);
        if (inv_tau) inv_tau[0] = it;
    }
}

int tau_grad(const float* q, const float* dqk, int lddq, const float* tau, int B, int D, float alpha, const float* scale_dev,
             float* dtau, float* inv_tau, hipStream_t st) {
    if (!q || !dqk || !tau || B <= 0 || D <= 0 || lddq < D || (!dtau && !inv_tau)) return SPN_ERR_ARG;
    hipLaunchKernelGGL(tau_grad_kernel, dim3(1), dim3(256), 0, st, q, dqk, lddq, tau, B, D, alpha, scale_dev, dtau, inv_tau);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// bf16 gradient exchange (distributed.GradBucketReducer(comm_dtype="bf16")): the ranks' bf16 chunks of one slice, stacked [G][m],
// summed in fp32 in RANK ORDER (identical on every rank) and rounded once to bf16 - the value every replica then receives.
__global__ __launch_bounds__(256) void sum_ranks_bf16_kernel(const bf16_t* __restrict__ x, int G, size_t m8, bf16_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m8; i += (size_t)gridDim.x * 256) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < G; ++r) {
            const bf16x8 v = *(const bf16x8*)(x + ((size_t)r * m8 + i) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
        *(bf16x8*)(out + i * 8) = o;
    }
}

int sum_ranks_bf16(const bf16_t* x, int G, size_t m, bf16_t* out, hipStream_t st) {
    if (!x || !out || G <= 0) return SPN_ERR_ARG;
    if (m % 8 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return SPN_ERR_SHAPE;
    if (!m) return SPN_OK;
    const size_t m8 = m / 8;
    const int blocks = (int)((m8 + 255) / 256 < 2048 ? (m8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(sum_ranks_bf16_kernel, dim3(blocks), dim3(256), 0, st, x, G, m8, out);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// fp32 flavour of the same exchange (direct reduce-scatter + all-gather over all xGMI links): out[i] = sum over the
// ranks in rank order of x[r][i]
__global__ __launch_bounds__(256) void sum_ranks_f32_kernel(const float* __restrict__ x, int G, size_t m4, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m4; i += (size_t)gridDim.x * 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < G; ++r) acc += *(const f32x4*)(x + ((size_t)r * m4 + i) * 4);
        *(f32x4*)(out + i * 4) = acc;
    }
}

int sum_ranks_f32(const float* x, int G, size_t m, float* out, hipStream_t st) {
    if (!x || !out || G <= 0) return SPN_ERR_ARG;
    if (m % 4 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return SPN_ERR_SHAPE;
    if (!m) return SPN_OK;
    const size_t m4 = m / 4;
    const int blocks = (int)((m4 + 255) / 256 < 2048 ? (m4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(sum_ranks_f32_kernel, dim3(blocks), dim3(256), 0, st, x, G, m4, out);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const bf16_t* s = x + i * 4;
        *(f32x4*)(y + i * 4) = f32x4{bf2f(s[0]), bf2f(s[1]), bf2f(s[2]), bf2f(s[3])};
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) y[i] = bf2f(x[i]);
}

int cast_bf16_f32(const bf16_t* x, float* y, size_t n, hipStream_t st) {
    if (!n) return SPN_OK;
    if (!x || !y) return SPN_ERR_ARG;
    if (((uintptr_t)y & 15) || ((uintptr_t)x & 7)) return SPN_ERR_SHAPE;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? ((n4 + 255) / 256 ? (n4 + 255) / 256 : 1) : 2048);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, st, x, y, n);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int cast_f32_bf16(const float* x, bf16_t* y, size_t n, hipStream_t st) {
    if (n == 0) return SPN_OK;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, st, x, y, n);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// y (optional, bf16 copy) and yt (optional, bf16 transpose) of x [rows, cols]; 32x32 tiles via LDS.
template <typename TIN>
__global__ void cast_transpose_kernel(const TIN* __restrict__ x, bf16_t* __restrict__ y, bf16_t* __restrict__ yt,
                                      int rows, int cols) {
    __shared__ float tile[32][33];
    const int tiles_c = (cols + 31) / 32;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = tr * 32 + ty + i * 8, c = tc * 32 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = (float)x[(size_t)r * cols + c];
            if (y) y[(size_t)r * cols + c] = f2bf(v);
        }
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
    if (yt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tc * 32 + ty + i * 8, r = tr * 32 + tx;
            if (r < rows && c < cols) yt[(size_t)c * rows + r] = f2bf(tile[tx][ty + i * 8]);
        }
    }
}

int cast_transpose_f32_bf16(const float* x, bf16_t* y, bf16_t* yt, int rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return SPN_ERR_ARG;
    const int tiles = ((rows + 31) / 32) * ((cols + 31) / 32);
    hipLaunchKernelGGL(cast_transpose_kernel<float>, dim3(tiles), dim3(256), 0, st, x, y, yt, rows, cols);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// The same tile routine over MANY matrices in one launch: `layers` repetitions (strides p_stride / wb_stride) of up
// to 4 matrices described by d; blockIdx.x -> (layer, matrix, tile).  One launch instead of 4 per layer for the
// bf16 weight refresh after every optimizer step.
__global__ void cast_transpose_multi_kernel(const float* __restrict__ p, bf16_t* __restrict__ wb, CastTransposeSet d) {
    __shared__ float tile[32][33];
    const int layer = blockIdx.x / d.tile_start[4];
    int t = blockIdx.x % d.tile_start[4];
    int m = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (t >= d.tile_start[i]) m = i;
    t -= d.tile_start[m];
    const int rows = d.rows[m], cols = d.cols[m];
    const float* x = p + (size_t)layer * d.p_stride + d.src[m];
    bf16_t* y = wb + (size_t)layer * d.wb_stride + d.dst[m];
    bf16_t* yt = wb + (size_t)layer * d.wb_stride + d.dst_t[m];
    const int tiles_c = (cols + 31) / 32;
    const int tr = t / tiles_c, tc = t % tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = tr * 32 + ty + i * 8, c = tc * 32 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = x[(size_t)r * cols + c];
            y[(size_t)r * cols + c] = f2bf(v);
        }
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tc * 32 + ty + i * 8, r = tr * 32 + tx;
        if (r < rows && c < cols) yt[(size_t)c * rows + r] = f2bf(tile[tx][ty + i * 8]);
    }
}

// 64x64 tiles for matrices whose sides are multiples of 64 (every weight of the towers): 16-byte loads, the bf16 copy
// written 8 bytes per lane, the transpose gathered from an LDS image of the bf16 tile and written 16 bytes per lane - all
// global accesses are whole 128 / 256-byte rows (the 32x32 routine above moves 4 / 2 bytes per lane in 64-byte rows:
// 202 us for ViT-L/14's 85 M block weights, 3.4 TB/s).
__global__ __launch_bounds__(256) void cast_transpose_multi64_kernel(const float* __restrict__ p, bf16_t* __restrict__ wb,
                                                                     CastTransposeSet d) {
    __shared__ bf16_t tile[64][66];
    const int layer = blockIdx.x / d.tile_start[4];
    int t = blockIdx.x % d.tile_start[4];
    int m = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (t >= d.tile_start[i]) m = i;
    t -= d.tile_start[m];
    const int rows = d.rows[m], cols = d.cols[m];
    const float* x = p + (size_t)layer * d.p_stride + d.src[m];
    bf16_t* y = wb + (size_t)layer * d.wb_stride + d.dst[m];
    bf16_t* yt = wb + (size_t)layer * d.wb_stride + d.dst_t[m];
    const int tiles_c = cols / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tid = threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(x + (size_t)(r0 + (tid >> 4) + i * 16) * cols + c0 + (tid & 15) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + i * 16, c = (tid & 15) * 4;
        const bf16x4 o = {f2bf(v[i][0]), f2bf(v[i][1]), f2bf(v[i][2]), f2bf(v[i][3])};
        *(bf16x4*)(y + (size_t)(r0 + r) * cols + c0 + c) = o;
        *(bf16x2*)(&tile[r][c]) = bf16x2{o[0], o[1]};
        *(bf16x2*)(&tile[r][c + 2]) = bf16x2{o[2], o[3]};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (tid >> 3) + i * 32, rr = (tid & 7) * 8;       // output row c (a source column), 8 source rows
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = tile[rr + k][c];
        *(bf16x8*)(yt + (size_t)(c0 + c) * rows + r0 + rr) = o;
    }
}

int cast_transpose_multi(const float* p, bf16_t* wb, const CastTransposeSet& d, int layers, hipStream_t st) {
    if (layers <= 0 || d.tile_start[4] <= 0) return SPN_ERR_ARG;
    bool all64 = true;
    CastTransposeSet d64 = d;
    for (int i = 0; i < 4; ++i) {
        const bool used = d.tile_start[i + 1] > d.tile_start[i];
        if (used && (d.rows[i] % 64 || d.cols[i] % 64 || d.dst[i] % 8 || d.dst_t[i] % 8 || d.src[i] % 4)) all64 = false;
        d64.tile_start[i + 1] = d64.tile_start[i] + (used ? (d.rows[i] / 64) * (d.cols[i] / 64) : 0);
    }
    if (all64 && d.p_stride % 4 == 0 && d.wb_stride % 8 == 0 && ((uintptr_t)p % 16) == 0 && ((uintptr_t)wb % 16) == 0) {
        hipLaunchKernelGGL(cast_transpose_multi64_kernel, dim3((unsigned)(layers * d64.tile_start[4])), dim3(256), 0, st, p, wb,
                           d64);
        SPN_CHECK_LAUNCH();
        return SPN_OK;
    }
    hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3((unsigned)(layers * d.tile_start[4])), dim3(256), 0, st, p, wb, d);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int transpose_bf16(const bf16_t* x, bf16_t* y, int rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return SPN_ERR_ARG;
    const int tiles = ((rows + 31) / 32) * ((cols + 31) / 32);
    hipLaunchKernelGGL(cast_transpose_kernel<bf16_t>, dim3(tiles), dim3(256), 0, st, x, (bf16_t*)nullptr, y, rows, cols);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------------ row fold
// out[c] = alpha * sum_{r<n} ws[r*stride + c] (+ out[c]).  Block = 16 column quads x 16 row
// lanes, so the n partial rows are read 16-wide in parallel instead of by one serial loop.
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ ws, size_t stride, int n, size_t C,
                                                        float* __restrict__ out, float alpha, int accumulate) {
    __shared__ f32x4 red[16][17];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const size_t c = ((size_t)blockIdx.x * 16 + cq) * 4;
    f32x4 s = {0, 0, 0, 0};
    if (c < C) {
#pragma unroll 4
        for (int r = rl; r < n; r += 16) s += *(const f32x4*)(ws + (size_t)r * stride + c);
    }
    red[rl][cq] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k][cq];
        s *= alpha;
        if (accumulate) s += *(const f32x4*)(out + c);
        *(f32x4*)(out + c) = s;
    }
}

// Zero fill of a gradient range that a scatter kernel then adds into (the token-embedding gradient: 152 MB for ViT-L/14).  The
// library's own kernel instead of hipMemsetAsync: one workgroup per CU walking 16-byte stores (the runtime's fill kernel runs at the
// same ~6 TB/s; this keeps every device activity of a step inside the library and on the caller's stream without a runtime hop).
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, size_t n4, size_t n) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) ((f32x4*)p)[i] = z;
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) p[i] = 0.f;
}

int zero_fill_f32(float* p, size_t n, hipStream_t st) {
    if (!n) return SPN_OK;
    if (!p) return SPN_ERR_ARG;
    size_t head = 0;
    if ((uintptr_t)p & 15) {                            // unaligned start: let the scalar tail path of a first tiny launch take it
        head = (16 - ((uintptr_t)p & 15)) / 4;
        if (head > n) head = n;
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, p, (size_t)0, head);
        SPN_CHECK_LAUNCH();
        p += head;
        n -= head;
        if (!n) return SPN_OK;
    }
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 ? (n4 + 255) / 256 : 1 : 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, st, p, n4, n);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int fold_rows(const float* ws, size_t stride, int n, size_t C, float* out, float alpha, int accumulate, hipStream_t st) {
    if (C % 4 || stride % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, st, ws, stride, n, C, out, alpha,
                       accumulate);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// The same for up to FOLD_BATCH_MAX (ws, out) pairs of one shape in ONE launch (grid.y = pair): the LayerNorm parameter
// gradients of a whole backward pass - 24 folds of 1024 x 1536 partials, each too small to fill the chip or to hide its
// own latency - leave in one go behind the grouped weight-gradient GEMM.
__global__ __launch_bounds__(256) void fold_rows_batched_kernel(const FoldBatch b) {
    __shared__ f32x4 red[16][17];
    const float* ws = b.ws[blockIdx.y];
    float* out = b.out[blockIdx.y];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const size_t c = ((size_t)blockIdx.x * 16 + cq) * 4;
    f32x4 s = {0, 0, 0, 0};
    if (c < b.C) {
#pragma unroll 4
        for (int r = rl; r < b.n; r += 16) s += *(const f32x4*)(ws + (size_t)r * b.stride + c);
    }
    red[rl][cq] = s;
    __syncthreads();
    if (rl == 0 && c < b.C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k][cq];
        *(f32x4*)(out + c) = s;
    }
}

int fold_rows_batched(const FoldBatch& b, hipStream_t st) {
    if (b.items <= 0 || b.items > FOLD_BATCH_MAX || b.C % 4 || b.stride % 4) return SPN_ERR_ARG;
    hipLaunchKernelGGL(fold_rows_batched_kernel, dim3((unsigned)((b.C + 63) / 64), b.items), dim3(256), 0, st, b);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ----------------------------------------------------------------------------- column sums
// out[c] (+)= sum_r x[r][c]; two stages: CS_ROWS row-slabs x column groups -> ws, then fold.
static constexpr int CS_SLABS = 64;

__global__ void colsum_partial_kernel(const bf16_t* __restrict__ x, int rows, int cols, int ld,
                                      float* __restrict__ ws) {
    // block: 256 threads = 32 column-octets x 8 row lanes; covers 256 columns
    const int cg = blockIdx.x, slab = blockIdx.y;
    const int co = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = cg * 256 + co * 8;
    const int rows_per = (rows + CS_SLABS - 1) / CS_SLABS;
    const int r0 = slab * rows_per, r1 = min(rows, r0 + rows_per);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < cols) {
        for (int r = r0 + rl; r < r1; r += 8) {
            const bf16x8 v = *(const bf16x8*)(x + (size_t)r * ld + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
    }
    __shared__ float red[8][256 + 8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][co * 8 + e] = acc[e];
    __syncthreads();
    const int t = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][t];
    if (cg * 256 + t < cols) ws[(size_t)slab * cols + cg * 256 + t] = s;
}

size_t colsum_workspace_bytes(int rows, int cols) { return (size_t)CS_SLABS * cols * sizeof(float); }

int colsum_bf16(const bf16_t* x, int rows, int cols, int ld, float* out, int accumulate, float* ws, size_t ws_bytes,
                hipStream_t st) {
    if (rows <= 0 || cols <= 0) return SPN_ERR_ARG;
    if (cols % 8 || ld % 8) return SPN_ERR_SHAPE;
    if (ws_bytes < colsum_workspace_bytes(rows, cols)) return SPN_ERR_WORKSPACE;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, CS_SLABS), dim3(256), 0, st, x, rows, cols, ld, ws);
    SPN_CHECK_LAUNCH();
    return fold_rows(ws, (size_t)cols, CS_SLABS, (size_t)cols, out, 1.0f, accumulate, st);
}

// ------------------------------------------------------------------------------- embedding
// x[b,l,:] = tok_emb[ids[b,l],:] + pos_emb[l,:]      (clip/model.py:346-348)
__global__ void embed_fwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ tok,
                                 const float* __restrict__ pos, float* __restrict__ x, int BL, int L, int W, int vocab) {
    const int w4 = W >> 2;
    const size_t total = (size_t)BL * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / w4), c = (int)(i % w4) * 4;
        int id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 t = *(const f32x4*)(tok + (size_t)id * W + c);
        const f32x4 p = *(const f32x4*)(pos + (size_t)(row % L) * W + c);
        *(f32x4*)(x + (size_t)row * W + c) = t + p;
    }
}

int embed_fwd(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* x, int B, int L, int W, int vocab,
              hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((size_t)B * L * (W / 4))), dim3(256), 0, st, ids, tok_emb,
                       pos_emb, x, B * L, L, W, vocab);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// dtok[ids[b,l],:] += dx[b,l,:] for l <= eot[b] (rows after the EOT token have zero gradient under
// the causal mask and are skipped); dtok must be zero on entry.  dpos[l,:] = sum_b dx[b,l,:].
// One wave per token row: dead rows (after the EOT token, or the skipped id) leave at once; a live row is added with
// lane-contiguous atomics (one instruction = 256 contiguous bytes; the earlier 4-floats-per-lane form spread each
// instruction over 16 cache lines: 73 us at B = 256, L = 77, this: see LABNOTES.md).
__global__ __launch_bounds__(256) void embed_bwd_tok_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ eot,
                                                           const float* __restrict__ dx, float* __restrict__ dtok, int BL,
                                                           int L, int W, int vocab, int skip_id) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < BL; row += gridDim.x * 4) {
        const int b = row / L, l = row - b * L;
        if (eot && l > eot[b]) continue;                    // wave-uniform
        int id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        if (id == skip_id) continue;
        const float* src = dx + (size_t)row * W;
        float* dst = dtok + (size_t)id * W;
        for (int c = lane; c < W; c += 64) atomicAdd(dst + c, src[c]);
    }
}

__global__ void embed_bwd_pos_kernel(const float* __restrict__ dx, float* __restrict__ dpos, int B, int L, int W) {
    const int l = blockIdx.x;
    for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
        f32x4 s = {0, 0, 0, 0};
        for (int b = 0; b < B; ++b) s += *(const f32x4*)(dx + ((size_t)b * L + l) * W + c);
        *(f32x4*)(dpos + (size_t)l * W + c) = s;
    }
}

int embed_bwd(const int32_t* ids, const int32_t* eot, const float* dx, float* dtok, float* dpos, int B, int L, int W,
              int vocab, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    if (dtok) {
        hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3(grid_for((size_t)B * L * 64, 256, 8192)), dim3(256), 0, st, ids, eot, dx,
                           dtok, B * L, L, W, vocab, -1);
        SPN_CHECK_LAUNCH();
    }
    if (dpos) {
        // dpos[l, :] = sum_b dx[b, l, :] is a fold of B "rows" of length L*W: the parallel fold kernel reads dx once
        // at streaming rate (the one-block-per-position loop over b took 70 us at B = 256)
        int rc = fold_rows(dx, (size_t)L * W, B, (size_t)L * W, dpos, 1.0f, 0, st);
        if (rc) return rc;
    }
    return SPN_OK;
}

// All rows live (TG-CIR's token output: the padding rows carry gradient too).  Most of the B*L rows then hold the SAME
// id (the zero padding), and their atomics on one dtok row serialise (1.4 ms at B = 256); those rows are summed by a
// slab reduction instead: partial[s, :] = sum of dx over the rows of slab s whose id is `hot_id`, folded into dtok[hot_id].
static constexpr int EB_SLAB = 64;

__global__ __launch_bounds__(256) void embed_bwd_hot_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dx,
                                                           float* __restrict__ part, int BL, int W, int hot_id) {
    const int r0 = blockIdx.x * EB_SLAB, r1 = min(BL, r0 + EB_SLAB);
    for (int c = threadIdx.x * 4; c < W; c += 256 * 4) {
        f32x4 s = {0, 0, 0, 0};
        for (int r = r0; r < r1; ++r)
            if (ids[r] == hot_id) s += *(const f32x4*)(dx + (size_t)r * W + c);
        *(f32x4*)(part + (size_t)blockIdx.x * W + c) = s;
    }
}

size_t embed_bwd_all_ws_bytes(int B, int L, int W) { return (size_t)((B * L + EB_SLAB - 1) / EB_SLAB) * W * sizeof(float); }

int embed_bwd_all(const int32_t* ids, const float* dx, float* dtok, float* dpos, int B, int L, int W, int vocab, int hot_id,
                  float* ws, size_t ws_bytes, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    if (ws_bytes < embed_bwd_all_ws_bytes(B, L, W)) return SPN_ERR_WORKSPACE;
    hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3(grid_for((size_t)B * L * 64, 256, 8192)), dim3(256), 0, st, ids, nullptr, dx, dtok,
                       B * L, L, W, vocab, hot_id);
    SPN_CHECK_LAUNCH();
    const int slabs = (B * L + EB_SLAB - 1) / EB_SLAB;
    hipLaunchKernelGGL(embed_bwd_hot_kernel, dim3(slabs), dim3(256), 0, st, ids, dx, ws, B * L, W, hot_id);
    SPN_CHECK_LAUNCH();
    int rc = fold_rows(ws, (size_t)W, slabs, (size_t)W, dtok + (size_t)hot_id * W, 1.0f, 0, st);
    if (rc) return rc;
    return fold_rows(dx, (size_t)L * W, B, (size_t)L * W, dpos, 1.0f, 0, st);
}

// eot[b] = argmax_l ids[b,l] (first maximum), clip/model.py:356
// one wave per caption: lanes take positions lane, lane+64, ...; (value, position) pairs are merged with the FIRST
// maximum winning (a serial loop per thread chained L dependent loads: 24 us at L = 77)
__global__ __launch_bounds__(256) void eot_argmax_kernel(const int32_t* __restrict__ ids, int32_t* __restrict__ eot, int B,
                                                        int L) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int best = INT32_MIN, bi = INT32_MAX;
    for (int l = lane; l < L; l += 64) {
        const int v = ids[(size_t)b * L + l];
        if (v > best) { best = v; bi = l; }              // ascending l within a lane: strict > keeps the first
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int v2 = __shfl_xor(best, off, 64), i2 = __shfl_xor(bi, off, 64);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (lane == 0) eot[b] = bi;
}

int eot_argmax(const int32_t* ids, int32_t* eot, int B, int L, hipStream_t st) {
    hipLaunchKernelGGL(eot_argmax_kernel, dim3((B + 3) / 4), dim3(256), 0, st, ids, eot, B, L);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ eot, float* __restrict__ out,
                                   int L, int W) {
    const int b = blockIdx.x;
    const float* src = x + ((size_t)b * L + eot[b]) * W;
    for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) *(f32x4*)(out + (size_t)b * W + c) = *(const f32x4*)(src + c);
}

int gather_rows_f32(const float* x, const int32_t* eot, float* out, int B, int L, int W, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, st, x, eot, out, L, W);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// dx[b,l,:] = (l == eot[b]) ? src[b,:] : 0 ; also the bf16 copy
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ eot, float* __restrict__ dx,
                                    bf16_t* __restrict__ dxb, int BL, int L, int W) {
    const int w4 = W >> 2;
    const size_t total = (size_t)BL * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / w4), c = (int)(i % w4) * 4;
        const int b = row / L, l = row % L;
        f32x4 v = {0, 0, 0, 0};
        if (l == eot[b]) v = *(const f32x4*)(src + (size_t)b * W + c);
        *(f32x4*)(dx + (size_t)row * W + c) = v;
        if (dxb) {
            bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)(dxb + (size_t)row * W + c) = o;
        }
    }
}

int scatter_rows_f32(const float* src, const int32_t* eot, float* dx, bf16_t* dx_bf16, int B, int L, int W,
                     hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((size_t)B * L * (W / 4))), dim3(256), 0, st, src, eot, dx,
                       dx_bf16, B * L, L, W);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------- packed sequences
// Dead-token elimination for the causal text tower: rows after the EOT token influence neither the pooled
// feature nor any gradient, so only the cu[B] live rows are materialised (sequence b = rows cu[b]..cu[b+1]-1).
__global__ void build_row_map_kernel(const int32_t* __restrict__ cu, int32_t* __restrict__ row_b,
                                     int32_t* __restrict__ row_l, int32_t* __restrict__ eot_row, int32_t* __restrict__ cu_copy) {
    const int b = blockIdx.x, r0 = cu[b], n = cu[b + 1] - r0;
    if (cu_copy && threadIdx.x == 0) {               // the caller's prefix sums into the activation arena (no runtime memcpy)
        cu_copy[b] = r0;
        if (b == (int)gridDim.x - 1) cu_copy[b + 1] = r0 + n;
    }
    for (int l = threadIdx.x; l < n; l += blockDim.x) {
        row_b[r0 + l] = b;
        row_l[r0 + l] = l;
    }
    if (threadIdx.x == 0) eot_row[b] = r0 + n - 1;
}

int build_row_map(const int32_t* cu, int32_t* row_b, int32_t* row_l, int32_t* eot_row, int B, hipStream_t st, int32_t* cu_copy) {
    hipLaunchKernelGGL(build_row_map_kernel, dim3(B), dim3(64), 0, st, cu, row_b, row_l, eot_row, cu_copy);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

__global__ void embed_fwd_packed_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ row_b,
                                        const int32_t* __restrict__ row_l, const float* __restrict__ tok,
                                        const float* __restrict__ pos, float* __restrict__ x, int T, int L, int W,
                                        int vocab) {
    const int w4 = W >> 2;
    const size_t total = (size_t)T * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / w4), c = (int)(i % w4) * 4;
        const int l = row_l[row];
        int id = ids[(size_t)row_b[row] * L + l];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        *(f32x4*)(x + (size_t)row * W + c) = *(const f32x4*)(tok + (size_t)id * W + c) + *(const f32x4*)(pos + (size_t)l * W + c);
    }
}

int embed_fwd_packed(const int32_t* ids, const int32_t* row_b, const int32_t* row_l, const float* tok_emb,
                     const float* pos_emb, float* x, int T, int L, int W, int vocab, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(embed_fwd_packed_kernel, dim3(grid_for((size_t)T * (W / 4))), dim3(256), 0, st, ids, row_b, row_l,
                       tok_emb, pos_emb, x, T, L, W, vocab);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// one wave per packed row, lane-contiguous atomics (see embed_bwd_tok_kernel)
__global__ __launch_bounds__(256) void embed_bwd_tok_packed_kernel(const int32_t* __restrict__ ids,
                                                                  const int32_t* __restrict__ row_b,
                                                                  const int32_t* __restrict__ row_l,
                                                                  const float* __restrict__ dx, float* __restrict__ dtok,
                                                                  int T, int L, int W, int vocab) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < T; row += gridDim.x * 4) {
        int id = ids[(size_t)row_b[row] * L + row_l[row]];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float* src = dx + (size_t)row * W;
        float* dst = dtok + (size_t)id * W;
        for (int c = lane; c < W; c += 64) atomicAdd(dst + c, src[c]);
    }
}

// dpos[l, :] = sum over the sequences that are longer than l of dx[cu[b] + l, :].  Workgroup (l, g) sums the sequences
// b = g, g + G, ... into part[g][l][:]; fold_rows adds the G partials (one workgroup per position looping over all B
// sequences took 76 us at B = 256).
static constexpr int EBP_GROUPS = 16;
__global__ __launch_bounds__(256) void embed_bwd_pos_packed_kernel(const float* __restrict__ dx, const int32_t* __restrict__ cu,
                                                                  float* __restrict__ part, int B, int L, int W) {
    const int l = blockIdx.x, g = blockIdx.y;
    for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
        f32x4 s = {0, 0, 0, 0};
        for (int b = g; b < B; b += EBP_GROUPS) {
            const int r0 = cu[b];
            if (l < cu[b + 1] - r0) s += *(const f32x4*)(dx + (size_t)(r0 + l) * W + c);
        }
        *(f32x4*)(part + ((size_t)g * L + l) * W + c) = s;
    }
}

size_t embed_bwd_packed_ws_bytes(int L, int W) { return (size_t)EBP_GROUPS * L * W * sizeof(float); }

int embed_bwd_packed(const int32_t* ids, const int32_t* row_b, const int32_t* row_l, const int32_t* cu, const float* dx,
                     float* dtok, float* dpos, int T, int B, int L, int W, int vocab, float* ws, size_t ws_bytes,
                     hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    if (ws_bytes < embed_bwd_packed_ws_bytes(L, W)) return SPN_ERR_WORKSPACE;
    hipLaunchKernelGGL(embed_bwd_tok_packed_kernel, dim3(grid_for((size_t)T * 64, 256, 8192)), dim3(256), 0, st, ids, row_b,
                       row_l, dx, dtok, T, L, W, vocab);
    SPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(embed_bwd_pos_packed_kernel, dim3(L, EBP_GROUPS), dim3(192), 0, st, dx, cu, ws, B, L, W);
    SPN_CHECK_LAUNCH();
    return fold_rows(ws, (size_t)L * W, EBP_GROUPS, (size_t)L * W, dpos, 1.0f, 0, st);
}

__global__ void gather_rows_abs_kernel(const float* __restrict__ x, const int32_t* __restrict__ rows,
                                       float* __restrict__ out, int W) {
    const int b = blockIdx.x;
    const float* src = x + (size_t)rows[b] * W;
    for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) *(f32x4*)(out + (size_t)b * W + c) = *(const f32x4*)(src + c);
}

int gather_rows_abs(const float* x, const int32_t* rows, float* out, int B, int W, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(gather_rows_abs_kernel, dim3(B), dim3(256), 0, st, x, rows, out, W);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

__global__ void scatter_rows_abs_kernel(const float* __restrict__ src, const int32_t* __restrict__ row_b,
                                        const int32_t* __restrict__ eot_row, float* __restrict__ dx,
                                        bf16_t* __restrict__ dxb, int T, int W) {
    const int w4 = W >> 2;
    const size_t total = (size_t)T * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / w4), c = (int)(i % w4) * 4;
        const int b = row_b[row];
        f32x4 v = {0, 0, 0, 0};
        if (row == eot_row[b]) v = *(const f32x4*)(src + (size_t)b * W + c);
        *(f32x4*)(dx + (size_t)row * W + c) = v;
        if (dxb) {
            bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)(dxb + (size_t)row * W + c) = o;
        }
    }
}

int scatter_rows_abs(const float* src, const int32_t* row_b, const int32_t* eot_row, float* dx, bf16_t* dx_bf16, int T,
                     int W, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(scatter_rows_abs_kernel, dim3(grid_for((size_t)T * (W / 4))), dim3(256), 0, st, src, row_b, eot_row,
                       dx, dx_bf16, T, W);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// Pooled rows of the text tower's last block (tower.hip: text_last_block_fwd / _bwd): sample b's pooled row is
// rows_abs[b] (packed layout) or b * L + eot[b] (dense).  gather: fp32 residual stream + bf16 attention output of those rows;
// scatter: the two gradients back to their rows, zeros everywhere else (the [T, W] buffers are written completely).
__global__ void gather_pool_rows_kernel(const float* __restrict__ x, const bf16_t* __restrict__ a, const int32_t* __restrict__ eot,
                                        const int32_t* __restrict__ rows_abs, int L, float* __restrict__ xo,
                                        bf16_t* __restrict__ ao, int W) {
    const int b = blockIdx.x;
    const size_t r = rows_abs ? (size_t)rows_abs[b] : (size_t)b * L + eot[b];
    for (int c = threadIdx.x * 4; c < W; c += blockDim.x * 4) {
        *(f32x4*)(xo + (size_t)b * W + c) = *(const f32x4*)(x + r * W + c);
        *(bf16x4*)(ao + (size_t)b * W + c) = *(const bf16x4*)(a + r * W + c);
    }
}

int gather_pool_rows(const float* x, const bf16_t* a, const int32_t* eot, const int32_t* rows_abs, int L, float* xo, bf16_t* ao,
                     int B, int W, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(gather_pool_rows_kernel, dim3(B), dim3(192), 0, st, x, a, eot, rows_abs, L, xo, ao, W);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

__global__ void scatter_pool_rows_kernel(const float* __restrict__ de, const bf16_t* __restrict__ da, const int32_t* __restrict__ eot,
                                         const int32_t* __restrict__ row_b, const int32_t* __restrict__ eot_row, int L,
                                         float* __restrict__ dx, bf16_t* __restrict__ dattn, int T, int W) {
    const int w4 = W >> 2;
    const size_t total = (size_t)T * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / w4), c = (int)(i % w4) * 4;
        const int b = row_b ? row_b[row] : row / L;
        const bool hit = row_b ? row == eot_row[b] : row - b * L == eot[b];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        bf16x4 g = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (hit) {
            v = *(const f32x4*)(de + (size_t)b * W + c);
            g = *(const bf16x4*)(da + (size_t)b * W + c);
        }
        *(f32x4*)(dx + (size_t)row * W + c) = v;
        *(bf16x4*)(dattn + (size_t)row * W + c) = g;
    }
}

int scatter_pool_rows(const float* de, const bf16_t* da, const int32_t* eot, const int32_t* row_b, const int32_t* eot_row, int L,
                      float* dx, bf16_t* dattn, int T, int W, hipStream_t st) {
    if (W % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(scatter_pool_rows_kernel, dim3(grid_for((size_t)T * (W / 4))), dim3(256), 0, st, de, da, eot, row_b, eot_row,
                       L, dx, dattn, T, W);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ---------------------------------------------------------------------------------- AdamW
// torch.optim.AdamW semantics (decoupled weight decay), train_negplus.py:77-83.  g is
// multiplied by inv_scale first (GradScaler unscale); the whole step is skipped when
// *found_inf != 0 (GradScaler.step semantics).
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float rsqrt_bc2, float inv_scale, const float* __restrict__ found_inf,
                             const float* __restrict__ grad_scale, const float* __restrict__ step_dev, int wt) {
    if (found_inf && *found_inf != 0.f) return;
    const __amdgpu_buffer_rsrc_t rs_p = wt_rsrc(p), rs_m = wt_rsrc(m), rs_v = wt_rsrc(v);
    if (grad_scale) inv_scale /= *grad_scale;            // GradScaler's scale, read on the device (no host sync)
    if (step_dev) {                                      // step count kept on the device (adamw_tick): skipped steps do not count
        const double t = (double)*step_dev;
        bc1 = (float)(1.0 - pow((double)b1, t));
        rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    const size_t n4 = n >> 2;
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 P = *(const f32x4*)(p + i * 4), G = *(const f32x4*)(g + i * 4);
        f32x4 Mv = *(const f32x4*)(m + i * 4), V = *(const f32x4*)(v + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = G[e] * inv_scale;
            float pe = P[e] * (1.0f - lr * wd);
            const float me = b1 * Mv[e] + (1.0f - b1) * gr;
            const float ve = b2 * V[e] + (1.0f - b2) * gr * gr;
            const float denom = sqrtf(ve) * rsqrt_bc2 + eps;
            pe -= step_size * (me / denom);
            P[e] = pe; Mv[e] = me; V[e] = ve;
        }
        if (wt) {                                        // write-through: the three output streams do not linger in L2
            store_wt16(rs_p, i * 16, __builtin_bit_cast(u32x4, P));
            store_wt16(rs_m, i * 16, __builtin_bit_cast(u32x4, Mv));
            store_wt16(rs_v, i * 16, __builtin_bit_cast(u32x4, V));
        } else {
            *(f32x4*)(p + i * 4) = P;
            *(f32x4*)(m + i * 4) = Mv;
            *(f32x4*)(v + i * 4) = V;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gr = g[i] * inv_scale;
        float pe = p[i] * (1.0f - lr * wd);
        const float me = b1 * m[i] + (1.0f - b1) * gr;
        const float ve = b2 * v[i] + (1.0f - b2) * gr * gr;
        pe -= step_size * (me / (sqrtf(ve) * rsqrt_bc2 + eps));
        p[i] = pe; m[i] = me; v[i] = ve;
    }
}

int adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
               float wd, int step, float inv_scale, const float* found_inf, hipStream_t st, const float* grad_scale,
               const float* step_dev) {
    if (n == 0) return SPN_OK;
    if (step < 1 && !step_dev) return SPN_ERR_ARG;
    if (step < 1) step = 1;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    // ONE workgroup per CU: the kernel streams seven arrays (3.5 GB for ViT-L/14's 124 M parameters) and is bound by HBM page
    // locality, not by bytes in flight - 2 048 workgroups: 4.5-4.9 TB/s, 256: 5.4-5.7 (a grid that is not a multiple of the
    // CU count loses 20 %: every workgroup does the same share in one round).  SPN_ADAMW_CAP overrides the grid.
    static const int cap = env_int_min1("SPN_ADAMW_CAP", device_cu_count());
    // SPN_ADAMW_WT=1: write-through stores of p / m / v - measured, no effect (13.34 vs 13.34 ms per step, kernel 605 vs 606 us): off
    static const int wt_env = [] { const char* e = spn_env("SPN_ADAMW_WT"); return e ? atoi(e) : 0; }();
    const int wt = (wt_env && (uint64_t)n * 4 < 0xfffffff0ull) ? 1 : 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1, 256, cap)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd,
                       (float)bc1, (float)(1.0 / sqrt(bc2)), inv_scale, found_inf, grad_scale, step_dev, wt);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// *step_dev += 1 unless the step is skipped (GradScaler.step does not call optimizer.step() on overflow, so torch's
// per-parameter `step` does not advance either).  Launched once per optimizer step, before the adamw_step launches.
__global__ void adamw_tick_kernel(float* __restrict__ step_dev, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.f) return;
    *step_dev += 1.0f;
}

int adamw_tick(float* step_dev, const float* found_inf, hipStream_t st) {
    if (!step_dev) return SPN_ERR_ARG;
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, st, step_dev, found_inf);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// GradScaler inf check: *found_inf = 1 if any g is non-finite (g is not modified; the
// unscale is folded into adamw_step through inv_scale).
__global__ void grad_check_kernel(const float* __restrict__ g, size_t n, float* __restrict__ found_inf) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = g[i];
        bad |= !(fabsf(x) <= 3.402823466e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;
}

int grad_unscale_check(float* g, size_t n, float inv_scale, float* found_inf, hipStream_t st) {
    (void)inv_scale;
    hipLaunchKernelGGL(grad_check_kernel, dim3(grid_for(n)), dim3(256), 0, st, g, n, found_inf);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
