// LayerNorm forward / backward (fp32 statistics, as clip/model.py:157-163 up-casts to fp32).
// One wave per row, the row held in registers (W <= 64*4*MAXV); HBM-bound streaming kernels.
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static constexpr int LN_MAXV_LIMIT = 8;   // float4 per lane -> W <= 2048

// y = (x - mean) * rstd * gamma + beta.  Measured at W = 768, 19 712 rows (inputs rotated through 480 MB): one row per wave
// and launch slot 17.8-19.3 us (4.7-5.1 TB/s); two rows per wave 21.3 us; the persistent walk below with 768 workgroups
// (3 per CU) 17.8 us, 256: 33 us, 512: 21.6, 1 024: 18.1, 2 048: 19.3.
// EXACT: W == 256 * LN_MAXV (768, 1024, ...): every lane owns LN_MAXV full column quads, no bounds tests (each one was
// an exec-mask branch around its load / store)
// ADD: the residual add of the block in front is fused in - the row is x + yin (yin bf16: the out-projection / c_proj GEMM
// result), the sum is written to xo (fp32 residual stream) and normalised.  yin may alias yb: a wave holds its whole row in
// registers before it stores anything, and rows belong to exactly one wave.
template <int LN_MAXV, bool EXACT, bool ADD = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* yb,
                                                            float* __restrict__ yf, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int W, float eps,
                                                            const bf16_t* yin = nullptr, float* __restrict__ xo = nullptr, int wt = 0) {
    const __amdgpu_buffer_rsrc_t rs_xo = wt_rsrc(xo), rs_yf = wt_rsrc(yf), rs_yb = wt_rsrc(yb);
    // persistent: wave w walks rows w, w + nwaves, ... with the next row's loads in flight under the current row's two
    // reductions and stores
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int nv = W >> 2;   // float4 count per row
    f32x4 gm[LN_MAXV], bt[LN_MAXV], v[LN_MAXV], nx[LN_MAXV];
    [[maybe_unused]] bf16x4 ya[LN_MAXV], ny[LN_MAXV];
    const bf16x4 zero4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + i * 64;
        gm[i] = (EXACT || c < nv) ? *(const f32x4*)(gamma + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        bt[i] = (EXACT || c < nv) ? *(const f32x4*)(beta + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        v[i] = (wave < rows && (EXACT || c < nv)) ? *(const f32x4*)(x + (size_t)wave * W + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (ADD) ya[i] = (wave < rows && (EXACT || c < nv)) ? *(const bf16x4*)(yin + (size_t)wave * W + c * 4) : zero4;
    }
    for (int row = wave; row < rows; row += nwaves) {
        const int nrow = row + nwaves;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            nx[i] = (nrow < rows && (EXACT || c < nv)) ? *(const f32x4*)(x + (size_t)nrow * W + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (ADD) ny[i] = (nrow < rows && (EXACT || c < nv)) ? *(const bf16x4*)(yin + (size_t)nrow * W + c * 4) : zero4;
        }
        if constexpr (ADD) {
#pragma unroll
            for (int i = 0; i < LN_MAXV; ++i) {
                const int c = lane + i * 64;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] += bf2f(ya[i][e]);
                if (EXACT || c < nv) {
                    if (wt) store_wt16(rs_xo, ((size_t)row * W + c * 4) * 4, __builtin_bit_cast(u32x4, v[i]));
                    else *(f32x4*)(xo + (size_t)row * W + c * 4) = v[i];
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];     // lanes past the row hold zeros
        const float mu = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (EXACT || c < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mu;
                    q += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)W + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (EXACT || c < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * gm[i][e] + bt[i][e];
                if (yf) {
                    if (wt) store_wt16(rs_yf, ((size_t)row * W + c * 4) * 4, __builtin_bit_cast(u32x4, o));
                    else *(f32x4*)(yf + (size_t)row * W + c * 4) = o;
                }
                if (yb) {
                    bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    if (wt) store_wt8(rs_yb, ((size_t)row * W + c * 4) * 2, __builtin_bit_cast(u32x2, ob));
                    else *(bf16x4*)(yb + (size_t)row * W + c * 4) = ob;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            v[i] = nx[i];
            if constexpr (ADD) ya[i] = ny[i];
        }
    }
}

int layernorm_fwd(const float* x, const float* gamma, const float* beta, bf16_t* y_bf16, float* y_f32, float* mean,
                  float* rstd, int rows, int W, float eps, hipStream_t st) {
    if (rows <= 0) return SPN_ERR_ARG;
    if (W % 4 || W > 64 * 4 * LN_MAXV_LIMIT) return SPN_ERR_SHAPE;
    static const int cap = env_int_min1("SPN_LNF_BLOCKS", 768);
    const int blocks = (rows + 3) / 4 < cap ? (rows + 3) / 4 : cap;
    const int wt = (spn_stream_wt() && (uint64_t)rows * W * 4 < 0xfffffff0ull) ? 1 : 0;
#define SPN_LN_FWD(V_)                                                                                          \
    do {                                                                                                        \
        if (W == 256 * V_)                                                                                      \
            hipLaunchKernelGGL((layernorm_fwd_kernel<V_, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y_bf16, y_f32, \
                               mean, rstd, rows, W, eps, (const bf16_t*)nullptr, (float*)nullptr, wt);          \
        else                                                                                                    \
            hipLaunchKernelGGL((layernorm_fwd_kernel<V_, false>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y_bf16,     \
                               y_f32, mean, rstd, rows, W, eps, (const bf16_t*)nullptr, (float*)nullptr, wt);   \
    } while (0)
    if (W <= 256) SPN_LN_FWD(1);
    else if (W <= 512) SPN_LN_FWD(2);
    else if (W <= 768) SPN_LN_FWD(3);
    else if (W <= 1024) SPN_LN_FWD(4);
    else SPN_LN_FWD(8);
#undef SPN_LN_FWD
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// x_out = x + y (y bf16, may alias y_bf16), then LayerNorm of x_out: the residual add of the GEMM in front fused into the
// normalisation that follows it (tower.hip: block_fwd with a deferred residual).  180 MB instead of 90 MB per call at
// [19 712, 768], but the GEMM in front stores 30 MB of bf16 instead of reading and writing 2 x 60 MB of fp32 in its epilogue.
int layernorm_fwd_add(const float* x, const bf16_t* y, const float* gamma, const float* beta, float* x_out, bf16_t* y_bf16,
                      float* mean, float* rstd, int rows, int W, float eps, hipStream_t st) {
    if (rows <= 0 || !x || !y || !x_out || !y_bf16) return SPN_ERR_ARG;
    if (W % 4 || W > 64 * 4 * LN_MAXV_LIMIT) return SPN_ERR_SHAPE;
    static const int cap = env_int_min1("SPN_LNF_BLOCKS", 768);
    const int blocks = (rows + 3) / 4 < cap ? (rows + 3) / 4 : cap;
    const int wt = (spn_stream_wt() && (uint64_t)rows * W * 4 < 0xfffffff0ull) ? 1 : 0;
#define SPN_LN_FWD_ADD(V_)                                                                                      \
    do {                                                                                                        \
        if (W == 256 * V_)                                                                                      \
            hipLaunchKernelGGL((layernorm_fwd_kernel<V_, true, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y_bf16, \
                               (float*)nullptr, mean, rstd, rows, W, eps, y, x_out, wt);                        \
        else                                                                                                    \
            hipLaunchKernelGGL((layernorm_fwd_kernel<V_, false, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y_bf16, \
                               (float*)nullptr, mean, rstd, rows, W, eps, y, x_out, wt);                        \
    } while (0)
    if (W <= 256) SPN_LN_FWD_ADD(1);
    else if (W <= 512) SPN_LN_FWD_ADD(2);
    else if (W <= 768) SPN_LN_FWD_ADD(3);
    else if (W <= 1024) SPN_LN_FWD_ADD(4);
    else SPN_LN_FWD_ADD(8);
#undef SPN_LN_FWD_ADD
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// Backward.  xhat = (x-mean)*rstd, g = dy*gamma:
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g*xhat));  dgamma = sum_r dy*xhat;  dbeta = sum_r dy
// Each wave walks rows wave_id, wave_id+nwaves, ... and keeps per-column dgamma/dbeta partials in
// registers; partials go to ws[nwaves_total][2][W] and are folded by a second kernel.
// grid of the backward kernel: ONE workgroup per CU (like AdamW, the kernel is bound by HBM page locality: at [19 712, 768]
// 256 workgroups 41.5 us, 512: 42.9, 1 024: 46.0, 2 048: 52.6 with the parameter-gradient fold); SPN_LNB_BLOCKS overrides

template <typename TDY, int LN_MAXV, bool EXACT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            int accumulate_dx, bf16_t* __restrict__ dxb,
                                                            float* __restrict__ ws, int rows, int W, int wt = 0) {
    const __amdgpu_buffer_rsrc_t rs_dx = wt_rsrc(dx), rs_dxb = wt_rsrc(dxb);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nv = W >> 2;
    f32x4 dg[LN_MAXV], db[LN_MAXV], gm[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        dg[i] = f32x4{0, 0, 0, 0};
        db[i] = f32x4{0, 0, 0, 0};
        const int c = lane + i * 64;
        gm[i] = (EXACT || c < nv) ? *(const f32x4*)(gamma + c * 4) : f32x4{0, 0, 0, 0};
    }
    for (int row = wave; row < rows; row += nwaves) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[LN_MAXV], g[LN_MAXV], dprev[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
        // the gradient already in dx (accumulate) is fetched together with x and dy, not after the row reduction
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            dprev[i] = (accumulate_dx && (EXACT || c < nv)) ? *(const f32x4*)(dx + (size_t)row * W + c * 4) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (EXACT || c < nv) {
                const f32x4 xv = *(const f32x4*)(x + (size_t)row * W + c * 4);
                f32x4 d;
                if constexpr (sizeof(TDY) == 2) {
                    const bf16x4 t = *(const bf16x4*)((const bf16_t*)dy + (size_t)row * W + c * 4);
                    d = f32x4{bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
                } else {
                    d = *(const f32x4*)((const float*)dy + (size_t)row * W + c * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    g[i][e] = d[e] * gm[i][e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                    dg[i][e] += d[e] * xh[i][e];
                    db[i][e] += d[e];
                }
            }
        }
        s1 = wave_sum(s1) / (float)W;
        s2 = wave_sum(s2) / (float)W;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (EXACT || c < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - s1 - xh[i][e] * s2);
                float* dp = dx + (size_t)row * W + c * 4;
                o += dprev[i];
                if (wt) store_wt16(rs_dx, ((size_t)row * W + c * 4) * 4, __builtin_bit_cast(u32x4, o));
                else *(f32x4*)dp = o;
                if (dxb) {
                    bf16x4 ob = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    if (wt) store_wt8(rs_dxb, ((size_t)row * W + c * 4) * 2, __builtin_bit_cast(u32x2, ob));
                    else *(bf16x4*)(dxb + (size_t)row * W + c * 4) = ob;
                }
            }
        }
    }
    if (ws) {
        // combine the block's 4 waves through LDS, one partial row [2W] per block
        extern __shared__ __attribute__((aligned(16))) float lnred[];
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + i * 64;
            if (EXACT || c < nv) {
                *(f32x4*)(lnred + (size_t)wv * 2 * W + c * 4) = dg[i];
                *(f32x4*)(lnred + (size_t)wv * 2 * W + W + c * 4) = db[i];
            }
        }
        __syncthreads();
        float* w0 = ws + (size_t)blockIdx.x * 2 * W;
        for (int c = threadIdx.x * 4; c < 2 * W; c += 1024) {
            f32x4 s = *(const f32x4*)(lnred + c);
#pragma unroll
            for (int k = 1; k < 4; ++k) s += *(const f32x4*)(lnred + (size_t)k * 2 * W + c);
            *(f32x4*)(w0 + c) = s;
        }
    }
}

static int lnb_blocks(int rows) {
    int b = (rows + 3) / 4;
    static const int cap = env_int_min1("SPN_LNB_BLOCKS", device_cu_count());
    return b > cap ? cap : b;
}

int layernorm_bwd_partial_rows(int rows) { return lnb_blocks(rows); }

size_t layernorm_bwd_workspace_bytes(int rows, int W) { return (size_t)lnb_blocks(rows) * 2 * W * sizeof(float); }

int layernorm_bwd(const bf16_t* dy_bf16, const float* dy_f32, const float* x, const float* gamma, const float* mean,
                  const float* rstd, float* dx, int accumulate_dx, bf16_t* dx_bf16, float* dgamma, float* dbeta,
                  int accumulate_dparam, int rows, int W, float* ws, size_t ws_bytes, hipStream_t st) {
    if (rows <= 0 || (!dy_bf16 && !dy_f32) || !dx) return SPN_ERR_ARG;
    if (W % 4 || W > 64 * 4 * LN_MAXV_LIMIT) return SPN_ERR_SHAPE;
    const bool want_param = dgamma && dbeta;
    if (want_param && ws_bytes < layernorm_bwd_workspace_bytes(rows, W)) return SPN_ERR_WORKSPACE;
    const int blocks = lnb_blocks(rows);
    const int wt = (spn_stream_wt() && (uint64_t)rows * W * 4 < 0xfffffff0ull) ? 1 : 0;
    float* wsp = want_param ? ws : nullptr;
    const size_t lds = want_param ? (size_t)4 * 2 * W * sizeof(float) : 0;
#define SPN_LN_BWD(V_)                                                                                           \
    do {                                                                                                         \
        if (dy_bf16 && W == 256 * V_)                                                                            \
            hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, V_, true>), dim3(blocks), dim3(256), lds, st, dy_bf16, x, gamma, \
                               mean, rstd, dx, accumulate_dx, dx_bf16, wsp, rows, W, wt);                            \
        else if (dy_bf16)                                                                                        \
            hipLaunchKernelGGL((layernorm_bwd_kernel<bf16_t, V_, false>), dim3(blocks), dim3(256), lds, st, dy_bf16, x, gamma, \
                               mean, rstd, dx, accumulate_dx, dx_bf16, wsp, rows, W, wt);                            \
        else if (W == 256 * V_)                                                                                  \
            hipLaunchKernelGGL((layernorm_bwd_kernel<float, V_, true>), dim3(blocks), dim3(256), lds, st, dy_f32, x, gamma,  \
                               mean, rstd, dx, accumulate_dx, dx_bf16, wsp, rows, W, wt);                            \
        else                                                                                                     \
            hipLaunchKernelGGL((layernorm_bwd_kernel<float, V_, false>), dim3(blocks), dim3(256), lds, st, dy_f32, x, gamma,  \
                               mean, rstd, dx, accumulate_dx, dx_bf16, wsp, rows, W, wt);                            \
    } while (0)
    if (W <= 256) SPN_LN_BWD(1);
    else if (W <= 512) SPN_LN_BWD(2);
    else if (W <= 768) SPN_LN_BWD(3);      // the text towers' width: 144 -> <= 128 registers
    else if (W <= 1024) SPN_LN_BWD(4);
    else SPN_LN_BWD(8);
#undef SPN_LN_BWD
    SPN_CHECK_LAUNCH();
    if (want_param && accumulate_dparam == 2) return SPN_OK;   // the caller folds ws [blocks][2W] itself (fold_rows_batched)
    if (want_param) {
        if (dbeta == dgamma + W)   // adjacent in the flat gradient buffer: one fold over [dgamma | dbeta]
            return fold_rows(ws, (size_t)2 * W, blocks, (size_t)2 * W, dgamma, 1.0f, accumulate_dparam, st);
        int rc = fold_rows(ws, (size_t)2 * W, blocks, (size_t)W, dgamma, 1.0f, accumulate_dparam, st);
        if (rc) return rc;
        rc = fold_rows(ws + W, (size_t)2 * W, blocks, (size_t)W, dbeta, 1.0f, accumulate_dparam, st);
        if (rc) return rc;
    }
    return SPN_OK;
}

}  // namespace spn
