// fp32-exact inference path of the two CLIP towers ("exact encode mode", SURVEY 7g ii): the same forward as
// tower.hip / vision.hip with every GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: an f32 fmaf chain, 1/16 of
// the bf16 rate), fp32 activations throughout and an fp32 attention.  For validation / retrieval, where the
// bf16 towers' ~1e-2 relative feature error flips near-ties of a Recall@K ranking; never used by the training step.
#include "tower.h"

namespace spn {

#define SPN_TRYX(x)                       \
    do {                                  \
        int rc__ = (x);                   \
        if (rc__ != SPN_OK) return rc__;  \
    } while (0)

static inline size_t alx(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------ fp32 GEMM
// C[M,N] = act(alpha * A[M,K] . op(B) + bias[N]) (+ resid[M,N]);  op(B) = B[N,K]^T (b_kn = 0) or B[K,N] (b_kn = 1).
// 128x128x16 tile, 4 waves (2x2) x (2x2) 32x32 MFMA tiles; operands staged through padded fp32 LDS tiles.
static constexpr int FB = 128, FK = 16, FP = FK + 1;

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int ACT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, int M,
                                                       int N, int K, int lda, int ldb, int b_kn,
                                                       const float* __restrict__ bias, const float* __restrict__ resid,
                                                       int ldr, float* __restrict__ C, int ldc, float alpha) {
    __shared__ float As[FB][FP], Bs[FB][FP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = (N + FB - 1) / FB;
    const int m0 = (blockIdx.x / tiles_n) * FB, n0 = (blockIdx.x % tiles_n) * FB;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += FK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {                        // A tile: 128 rows x 16 k
            const int r = (tid >> 2) + 64 * it, c = (tid & 3) * 4;
            const int gm = m0 + r, gk = k0 + c;
            f32x4 v = {0, 0, 0, 0};
            if (gm < M) {
                const float* p = A + (size_t)gm * lda + gk;
                if (gk + 3 < K) v = *(const f32x4*)p;
                else
                    for (int e = 0; e < 4; ++e) v[e] = gk + e < K ? p[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) As[r][c + e] = v[e];
        }
        if (!b_kn) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {                    // B tile from [N, K]
                const int r = (tid >> 2) + 64 * it, c = (tid & 3) * 4;
                const int gn = n0 + r, gk = k0 + c;
                f32x4 v = {0, 0, 0, 0};
                if (gn < N) {
                    const float* p = B + (size_t)gn * ldb + gk;
                    if (gk + 3 < K) v = *(const f32x4*)p;
                    else
                        for (int e = 0; e < 4; ++e) v[e] = gk + e < K ? p[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[r][c + e] = v[e];
            }
        } else {
#pragma unroll
            for (int it = 0; it < 2; ++it) {                    // B tile from [K, N]: 16 k rows x 128 columns
                const int kr = (tid >> 5) + 8 * it, c = (tid & 31) * 4;
                const int gk = k0 + kr, gn = n0 + c;
                f32x4 v = {0, 0, 0, 0};
                if (gk < K) {
                    const float* p = B + (size_t)gk * ldb + gn;
                    if (gn + 3 < N) v = *(const f32x4*)p;
                    else
                        for (int e = 0; e < 4; ++e) v[e] = gn + e < N ? p[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[c + e][kr] = v[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FK / 2; ++kk) {
            const int kc = kk * 2 + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[wr * 64 + i * 32 + (lane & 31)][kc];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[wc * 64 + j * 32 + (lane & 31)][kc];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f32(b[j], a[i], acc[i][j]);   // swapped: lane owns row m
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wr * 64 + i * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= N) continue;
                    float v = acc[i][j][4 * g + e] * alpha;
                    if (bias) v += bias[n + e];
                    if (ACT == ACT_QUICKGELU) v = v / (1.0f + expf(-1.702f * v));
                    else if (ACT == ACT_GELU_ERF) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
                    if (resid) v += resid[(size_t)m * ldr + n + e];
                    if (ACT == ACT_RELU_POST) v = fmaxf(v, 0.f);
                    C[(size_t)m * ldc + n + e] = v;
                }
            }
    }
}

int gemm_f32(const float* A, const float* B, int M, int N, int K, int lda, int ldb, int b_kn, const float* bias, int act,
             const float* resid, int ldr, float* C, int ldc, float alpha, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return SPN_ERR_ARG;
    if (lda % 4 || ldb % 4 || ((uintptr_t)A | (uintptr_t)B) % 16) return SPN_ERR_SHAPE;
    const dim3 grid(((M + FB - 1) / FB) * ((N + FB - 1) / FB));
    if (act == ACT_NONE)
        hipLaunchKernelGGL(gemm_f32_kernel<ACT_NONE>, grid, dim3(256), 0, st, A, B, M, N, K, lda, ldb, b_kn, bias, resid, ldr, C, ldc, alpha);
    else if (act == ACT_QUICKGELU)
        hipLaunchKernelGGL(gemm_f32_kernel<ACT_QUICKGELU>, grid, dim3(256), 0, st, A, B, M, N, K, lda, ldb, b_kn, bias, resid, ldr, C, ldc, alpha);
    else if (act == ACT_GELU_ERF)
        hipLaunchKernelGGL(gemm_f32_kernel<ACT_GELU_ERF>, grid, dim3(256), 0, st, A, B, M, N, K, lda, ldb, b_kn, bias, resid, ldr, C, ldc, alpha);
    else if (act == ACT_RELU_POST)
        hipLaunchKernelGGL(gemm_f32_kernel<ACT_RELU_POST>, grid, dim3(256), 0, st, A, B, M, N, K, lda, ldb, b_kn, bias, resid, ldr, C, ldc, alpha);
    else
        return SPN_ERR_ARG;
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// --------------------------------------------------------------------------- ModifiedResNet pieces
// CLIP's ResNet image towers (clip/model.py:10-155: RN50, RN101, RN50x4 - the argparse default of
// train_negplus.py:192) run on this fp32 path only: NHWC activations, every convolution = im2col + gemm_f32 with the
// (eval-mode) BatchNorm folded into weight and bias on the host, AvgPool2d, and the single-query attention pool.
// out[(b, oy, ox), c*9 + ky*3 + kx] = x[b, oy*s + ky - 1, ox*s + kx - 1, c] (zero padding 1); columns >= 9C are zero.
__global__ void im2col3x3_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int C,
                                     int stride, int nchw, int Ho, int Wo, int ldk) {
    const size_t total = (size_t)B * Ho * Wo * ldk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % ldk);
        const size_t row = i / ldk;
        float v = 0.f;
        if (col < 9 * C) {
            const int c = col / 9, ky = (col % 9) / 3, kx = col % 3;
            const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), b = (int)(row / ((size_t)Wo * Ho));
            const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = nchw ? x[(((size_t)b * C + c) * H + iy) * W + ix] : x[(((size_t)b * H + iy) * W + ix) * C + c];
        }
        out[i] = v;
    }
}

int im2col3x3_f32(const float* x, float* out, int B, int H, int W, int C, int stride, int nchw, int ldk, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || stride <= 0 || ldk < 9 * C) return SPN_ERR_ARG;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const size_t n = (size_t)B * Ho * Wo * ldk;
    hipLaunchKernelGGL(im2col3x3_f32_kernel, dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, st,
                       x, out, B, H, W, C, stride, nchw, Ho, Wo, ldk);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// nn.AvgPool2d(k) (kernel = stride = k, no padding) on NHWC
__global__ void avgpool_nhwc_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const size_t total = (size_t)B * Ho * Wo * C;
    const float inv = 1.0f / (float)(k * k);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += x[(((size_t)b * H + oy * k + dy) * W + ox * k + dx) * C + c];
        y[i] = s * inv;
    }
}

int avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int k, hipStream_t st) {
    if (B <= 0 || C <= 0 || k <= 0 || H < k || W < k) return SPN_ERR_ARG;
    const size_t n = (size_t)B * (H / k) * (W / k) * C;
    hipLaunchKernelGGL(avgpool_nhwc_f32_kernel, dim3((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256)), dim3(256), 0, st,
                       x, y, B, H, W, C, k);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// AttentionPool2d token assembly (clip/model.py:69-71): tok[b,0] = mean_i x[b,i] + pos[0]; tok[b,1+i] = x[b,i] + pos[1+i]
__global__ void attnpool_tokens_f32_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ tok,
                                           int B, int HW, int C) {
    const size_t total = (size_t)B * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), b = (int)(i / C);
        float s = 0.f;
        for (int j = 0; j < HW; ++j) {
            const float v = x[((size_t)b * HW + j) * C + c];
            s += v;
            tok[((size_t)b * (HW + 1) + 1 + j) * C + c] = v + pos[(size_t)(1 + j) * C + c];
        }
        tok[(size_t)b * (HW + 1) * C + c] = s / (float)HW + pos[c];
    }
}

int attnpool_tokens_f32(const float* x, const float* pos, float* tok, int B, int HW, int C, hipStream_t st) {
    if (B <= 0 || HW <= 0 || C <= 0) return SPN_ERR_ARG;
    const size_t n = (size_t)B * C;
    hipLaunchKernelGGL(attnpool_tokens_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, pos, tok, B, HW, C);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// single-query multi-head attention (the pooled token attends to all S tokens), head_dim 64: one wave per (b, h)
__global__ void attnpool_attend_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                           float* __restrict__ out, int B, int S, int H) {
    const int lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= B * H) return;
    const int b = bh / H, h = bh % H, C = H * 64;
    const float qd = q[(size_t)b * C + h * 64 + lane] * 0.125f;
    float mx = -INFINITY, l = 0.f, o = 0.f;
    for (int j = 0; j < S; ++j) {
        const size_t row = ((size_t)b * S + j) * C + h * 64 + lane;
        const float s = wave_sum(qd * k[row]);
        const float mn = fmaxf(mx, s);
        const float corr = expf(mx - mn), p = expf(s - mn);
        l = l * corr + p;
        o = o * corr + p * v[row];
        mx = mn;
    }
    out[(size_t)b * C + h * 64 + lane] = o / l;
}

int attnpool_attend_f32(const float* q, const float* k, const float* v, float* out, int B, int S, int H, hipStream_t st) {
    if (B <= 0 || S <= 0 || H <= 0) return SPN_ERR_ARG;
    hipLaunchKernelGGL(attnpool_attend_f32_kernel, dim3((B * H + 3) / 4), dim3(256), 0, st, q, k, v, out, B, S, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// --------------------------------------------------------------------------- fp32 attention
// One thread per query row (q and the running output in registers), keys / values streamed through LDS in tiles of
// 64 rows, online softmax.  qkv: fp32 [B*L, 3W] (q | k | v, head h at column h*64); out fp32 [B*L, W].
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B,
                                                            int H, int L, int causal, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[64][64], Vs[64][64];
    const int W = H * 64, ld = 3 * W;
    const int nqc = (L + 255) / 256;
    const int qc = blockIdx.x % nqc, h = (blockIdx.x / nqc) % H, b = blockIdx.x / (nqc * H);
    const int qi = qc * 256 + threadIdx.x;
    const bool qok = qi < L;
    float q[64], o[64];
    const float* qp = qkv + ((size_t)b * L + (qok ? qi : 0)) * ld + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        const f32x4 v = *(const f32x4*)(qp + d);
#pragma unroll
        for (int e = 0; e < 4; ++e) { q[d + e] = v[e] * scale; o[d + e] = 0.f; }
    }
    float mx = -INFINITY, l = 0.f;
    const int kend = causal ? min(L, qc * 256 + 256) : L;
    for (int j0 = 0; j0 < kend; j0 += 64) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = (threadIdx.x >> 4) + 16 * it, c = (threadIdx.x & 15) * 4;
            f32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
            if (j0 + r < L) {
                const float* p = qkv + ((size_t)b * L + j0 + r) * ld + h * 64 + c;
                kv = *(const f32x4*)(p + W);
                vv = *(const f32x4*)(p + 2 * W);
            }
            *(f32x4*)&Ks[r][c] = kv;
            *(f32x4*)&Vs[r][c] = vv;
        }
        __syncthreads();
        const int jn = min(64, kend - j0);
        for (int j = 0; j < jn; ++j) {
            if (causal && j0 + j > qi) break;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 64; d += 4) {
                const f32x4 kv = *(const f32x4*)&Ks[j][d];
                s += q[d] * kv[0] + q[d + 1] * kv[1] + q[d + 2] * kv[2] + q[d + 3] * kv[3];
            }
            const float mn = fmaxf(mx, s);
            const float corr = expf(mx - mn), p = expf(s - mn);
            l = l * corr + p;
            mx = mn;
#pragma unroll
            for (int d = 0; d < 64; d += 4) {
                const f32x4 vv = *(const f32x4*)&Vs[j][d];
                o[d] = o[d] * corr + p * vv[0];
                o[d + 1] = o[d + 1] * corr + p * vv[1];
                o[d + 2] = o[d + 2] * corr + p * vv[2];
                o[d + 3] = o[d + 3] * corr + p * vv[3];
            }
        }
    }
    if (!qok) return;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float* op = out + ((size_t)b * L + qi) * W + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d += 4) *(f32x4*)(op + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
}

static int attention_f32(const float* qkv, float* out, int B, int H, int L, int causal, hipStream_t st) {
    const int nqc = (L + 255) / 256;
    hipLaunchKernelGGL(attention_f32_kernel, dim3(B * H * nqc), dim3(256), 0, st, qkv, out, B, H, L, causal, 0.125f);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------------ one block
// scratch (all fp32): h [T,W] | qkv [T,3W] | attn [T,W] | x_mid [T,W] | u [T,4W]
static int block_fwd_exact(const BlockCfg& c, const float* p, const float* x_in, float* x_out, float* scr, hipStream_t st) {
    int64_t o[13];
    block_param_offsets(c.W, o);
    const int T = c.B * c.L, W = c.W;
    float* h = scr;
    float* qkv = h + (size_t)T * W;
    float* attn = qkv + (size_t)T * 3 * W;
    float* x_mid = attn + (size_t)T * W;
    float* u = x_mid + (size_t)T * W;
    SPN_TRYX(layernorm_fwd(x_in, p + o[0], p + o[1], nullptr, h, nullptr, nullptr, T, W, c.eps, st));
    SPN_TRYX(gemm_f32(h, p + o[2], T, 3 * W, W, W, W, 0, p + o[3], ACT_NONE, nullptr, 0, qkv, 3 * W, 1.0f, st));
    SPN_TRYX(attention_f32(qkv, attn, c.B, c.H, c.L, c.causal, st));
    SPN_TRYX(gemm_f32(attn, p + o[4], T, W, W, W, W, 0, p + o[5], ACT_NONE, x_in, W, x_mid, W, 1.0f, st));
    SPN_TRYX(layernorm_fwd(x_mid, p + o[6], p + o[7], nullptr, h, nullptr, nullptr, T, W, c.eps, st));
    SPN_TRYX(gemm_f32(h, p + o[8], T, 4 * W, W, W, W, 0, p + o[9], c.act, nullptr, 0, u, 4 * W, 1.0f, st));
    SPN_TRYX(gemm_f32(u, p + o[10], T, W, 4 * W, 4 * W, 4 * W, 0, p + o[11], ACT_NONE, x_mid, W, x_out, W, 1.0f, st));
    return SPN_OK;
}

// ------------------------------------------------------------------------------ text tower
size_t text_exact_ws_bytes(const TextCfg& c) {
    const size_t T = (size_t)c.B * c.L;
    return alx((size_t)c.B * 4) + 2 * alx(T * c.W * 4) + alx(T * c.W * 4 * 10) + 2 * alx((size_t)c.B * c.W * 4);
}

int text_fwd_exact(const TextCfg& c, const float* params, const int32_t* ids, char* ws, size_t ws_bytes, float* feats,
                   hipStream_t st) {
    if (c.B <= 0 || c.L <= 0 || c.L > c.L_ctx || c.layers <= 0 || c.T != 0) return SPN_ERR_ARG;
    if (c.W % 64 || c.H * 64 != c.W || c.D % 4) return SPN_ERR_SHAPE;
    if (ws_bytes < text_exact_ws_bytes(c)) return SPN_ERR_WORKSPACE;
    TextLayout t;
    text_layout(c, &t);
    BlockCfg bc;
    bc.B = c.B; bc.L = c.L; bc.W = c.W; bc.H = c.H; bc.causal = 1; bc.act = ACT_QUICKGELU; bc.eps = 1e-5f;
    const size_t T = (size_t)c.B * c.L;
    char* p = ws;
    auto take = [&](size_t bytes) { char* r = p; p += alx(bytes); return r; };
    int32_t* eot = (int32_t*)take((size_t)c.B * 4);
    float* xa = (float*)take(T * c.W * 4);
    float* xb = (float*)take(T * c.W * 4);
    float* scr = (float*)take(T * c.W * 4 * 10);
    float* e = (float*)take((size_t)c.B * c.W * 4);
    float* le = (float*)take((size_t)c.B * c.W * 4);
    SPN_TRYX(eot_argmax(ids, eot, c.B, c.L, st));
    SPN_TRYX(embed_fwd(ids, params + t.tok, params + t.pos, xa, c.B, c.L, c.W, c.vocab, st));
    float *cur = xa, *nxt = xb;
    for (int l = 0; l < c.layers; ++l) {
        SPN_TRYX(block_fwd_exact(bc, params + t.blocks + t.block_size * l, cur, nxt, scr, st));
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    SPN_TRYX(gather_rows_f32(cur, eot, e, c.B, c.L, c.W, st));
    SPN_TRYX(layernorm_fwd(e, params + t.lnf_g, params + t.lnf_b, nullptr, le, nullptr, nullptr, c.B, c.W, 1e-5f, st));
    return gemm_f32(le, params + t.text_proj, c.B, c.D, c.W, c.W, c.D, 1, nullptr, ACT_NONE, nullptr, 0, feats, c.D, 1.0f, st);
}

// ---------------------------------------------------------------------------- vision tower
__global__ void im2col_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int R, int p) {
    const int g = R / p, K = 3 * p * p;
    const size_t total = (size_t)B * g * g * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % K);
        const size_t row = i / K;
        const int ch = col / (p * p), ky = (col / p) % p, kx = col % p;
        const int gx = (int)(row % g), gy = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
        out[i] = img[(((size_t)b * 3 + ch) * R + gy * p + ky) * R + gx * p + kx];
    }
}

__global__ void assemble_tokens_f32_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                           const float* __restrict__ pos, float* __restrict__ x, int B, int S, int W) {
    const size_t total = (size_t)B * S * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const size_t row = i / W;
        const int s = (int)(row % S), b = (int)(row / S);
        x[i] = (s == 0 ? cls[c] : emb[((size_t)b * (S - 1) + s - 1) * W + c]) + pos[(size_t)s * W + c];
    }
}

static int grid1x(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

size_t vision_exact_ws_bytes(const VisionCfg& c) {
    const int g = c.res / c.patch, S = g * g + 1;
    const size_t T = (size_t)c.B * S, P = (size_t)c.B * (S - 1), K = 3ull * c.patch * c.patch;
    const size_t Kq = (K + 3) & ~(size_t)3;
    return alx(P * Kq * 4) + alx(P * c.W * 4) + 3 * alx(T * c.W * 4) + alx(T * c.W * 4 * 10) + alx((size_t)c.B * 4) +
           2 * alx((size_t)c.B * c.W * 4);
}

int vision_fwd_exact(const VisionCfg& c, const float* params, const float* image, char* ws, size_t ws_bytes, float* feats,
                     hipStream_t st) {
    if (c.B <= 0 || c.layers <= 0 || c.patch <= 0 || c.res % c.patch || c.kind != 0) return SPN_ERR_ARG;
    if (c.W % 64 || c.H * 64 != c.W || c.D % 4 || (3 * c.patch * c.patch) % 4) return SPN_ERR_SHAPE;
    if (ws_bytes < vision_exact_ws_bytes(c)) return SPN_ERR_WORKSPACE;
    VisionLayout t;
    vision_layout(c, &t);
    const int g = c.res / c.patch, S = g * g + 1, K = 3 * c.patch * c.patch;
    const size_t T = (size_t)c.B * S, P = (size_t)c.B * (S - 1);
    BlockCfg bc;
    bc.B = c.B; bc.L = S; bc.W = c.W; bc.H = c.H; bc.causal = 0; bc.act = ACT_QUICKGELU; bc.eps = 1e-5f;
    char* p = ws;
    auto take = [&](size_t bytes) { char* r = p; p += alx(bytes); return r; };
    float* patches = (float*)take(P * K * 4);
    float* emb = (float*)take(P * c.W * 4);
    float* tok = (float*)take(T * c.W * 4);
    float* xa = (float*)take(T * c.W * 4);
    float* xb = (float*)take(T * c.W * 4);
    float* scr = (float*)take(T * c.W * 4 * 10);
    int32_t* zero_idx = (int32_t*)take((size_t)c.B * 4);
    float* cls_rows = (float*)take((size_t)c.B * c.W * 4);
    float* ln_cls = (float*)take((size_t)c.B * c.W * 4);
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(grid1x(P * K)), dim3(256), 0, st, image, patches, c.B, c.res, c.patch);
    SPN_CHECK_LAUNCH();
    SPN_TRYX(gemm_f32(patches, params + t.conv1, (int)P, c.W, K, K, K, 0, nullptr, ACT_NONE, nullptr, 0, emb, c.W, 1.0f, st));
    hipLaunchKernelGGL(assemble_tokens_f32_kernel, dim3(grid1x(T * c.W)), dim3(256), 0, st, emb, params + t.cls,
                       params + t.pos, tok, c.B, S, c.W);
    SPN_CHECK_LAUNCH();
    SPN_TRYX(layernorm_fwd(tok, params + t.ln_pre_g, params + t.ln_pre_b, nullptr, xa, nullptr, nullptr, (int)T, c.W, 1e-5f, st));
    float *cur = xa, *nxt = xb;
    for (int l = 0; l < c.layers; ++l) {
        SPN_TRYX(block_fwd_exact(bc, params + t.blocks + t.block_size * l, cur, nxt, scr, st));
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    hipError_t he = hipMemsetAsync(zero_idx, 0, (size_t)c.B * 4, st);
    if (he != hipSuccess) return (int)he;
    SPN_TRYX(gather_rows_f32(cur, zero_idx, cls_rows, c.B, S, c.W, st));
    SPN_TRYX(layernorm_fwd(cls_rows, params + t.ln_post_g, params + t.ln_post_b, nullptr, ln_cls, nullptr, nullptr, c.B, c.W,
                           1e-5f, st));
    return gemm_f32(ln_cls, params + t.proj, c.B, c.D, c.W, c.W, c.D, 1, nullptr, ACT_NONE, nullptr, 0, feats, c.D, 1.0f, st);
}

}  // namespace spn
