// TG-CIR second-stage head: the query producer of tgcir/models.py that sits between the CLIP text
// tower and the bank InfoNCE kernels.
//
//   Backbone.extract_text_fea (tgcir/models.py:127-151): tokens = ln_final(x) [B, L, C]; global_fea = pooled @ proj;
//       global tokens  g_i = global_fea * relu(masks_text[i])                     i < G (= 4)
//       local tokens   z = text_fc(tokens); a_s = sigmoid(conv1d_s(z)) [B, L];    s < S (= 8)
//                      l_s = mean_l (z * a_s)                                      (TokenLearner / SpatialAttention, :21-49)
//       mod_token = cat(g, l) [B, G + S, C]
//   CIRPlus.img_txt_fusion (:198-205): r = sigmoid(W2 relu(W1 cat(ref, mod) + b1) + b2) [B, NT, 1];
//       fuse = r * ref + (1 - r) * mod; q = normalize(mean_t fuse)
//
// The two Linear layers with a real GEMM shape (text_fc over B*L rows, s_remain_map[0] over B*NT rows) run on the bf16
// MFMA GEMMs (gemm_nt / gemm_tn); everything here is the per-sample glue around them, one block per sample, fp32.
// B*NT*C is ~1.5 M elements at B = 256: these kernels are launch-sized, far below every roofline that matters.
#include "common.h"
#include "kernels.h"

#define SPN_TRYG(x)                 \
    do {                            \
        const int rc_ = (x);        \
        if (rc_ != SPN_OK) return rc_; \
    } while (0)

namespace spn {

static constexpr int TG_S = 8;        // local tokens (TokenLearner heads)
static constexpr int TG_MAXL = 640;   // positions per sample (77 text tokens; 197 / 257 / 577 image tokens)
static constexpr int TG_MAXNT = 16;   // tokens per sample

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------- TokenLearner
// A[b, l, s] = sigmoid(<z[b, l, :], w[s, :]> + bias[s]);  mod[b, G + s, :] = (1/L) sum_l A[b, l, s] z[b, l, :]
__global__ __launch_bounds__(256) void tg_tokenlearn_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ A,
                                                               float* __restrict__ mod, int L, int C, int G) {
    __shared__ float sA[TG_MAXL][TG_S];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* zb = z + (size_t)b * L * C;
    for (int l = wid; l < L; l += 4) {
        float p[TG_S];
#pragma unroll
        for (int s = 0; s < TG_S; ++s) p[s] = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float zv = zb[(size_t)l * C + c];
#pragma unroll
            for (int s = 0; s < TG_S; ++s) p[s] += zv * w[s * C + c];
        }
#pragma unroll
        for (int s = 0; s < TG_S; ++s) {
            const float t = wave_sum(p[s]);
            if (lane == 0) {
                const float a = sigmoidf_(t + bias[s]);
                sA[l][s] = a;
                A[((size_t)b * L + l) * TG_S + s] = a;
            }
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)L;
    for (int c = tid; c < C; c += 256) {
        float acc[TG_S];
#pragma unroll
        for (int s = 0; s < TG_S; ++s) acc[s] = 0.f;
        for (int l = 0; l < L; ++l) {
            const float zv = zb[(size_t)l * C + c];
#pragma unroll
            for (int s = 0; s < TG_S; ++s) acc[s] += sA[l][s] * zv;
        }
#pragma unroll
        for (int s = 0; s < TG_S; ++s) mod[((size_t)b * (G + TG_S) + G + s) * C + c] = acc[s] * inv;
    }
}

// dA = (1/L) <dl_s, z_l>; P = dA A (1 - A);  dz[l] = sum_s (A[l,s]/L) dl_s + P[l,s] w_s;  dw_s = sum_l P[l,s] z_l
__global__ __launch_bounds__(256) void tg_tokenlearn_bwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                               const float* __restrict__ A, const float* __restrict__ dmod,
                                                               bf16_t* __restrict__ dz, float* __restrict__ dwpart,
                                                               float* __restrict__ dbpart, int L, int C, int G) {
    __shared__ float sA[TG_MAXL][TG_S], sP[TG_MAXL][TG_S];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* zb = z + (size_t)b * L * C;
    const float* dl = dmod + ((size_t)b * (G + TG_S) + G) * C;      // [S][C]
    const float inv = 1.0f / (float)L;
    for (int l = wid; l < L; l += 4) {
        float p[TG_S];
#pragma unroll
        for (int s = 0; s < TG_S; ++s) p[s] = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float zv = zb[(size_t)l * C + c];
#pragma unroll
            for (int s = 0; s < TG_S; ++s) p[s] += zv * dl[s * C + c];
        }
#pragma unroll
        for (int s = 0; s < TG_S; ++s) {
            const float t = wave_sum(p[s]) * inv;
            if (lane == 0) {
                const float a = A[((size_t)b * L + l) * TG_S + s];
                sA[l][s] = a;
                sP[l][s] = t * a * (1.0f - a);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float dls[TG_S], ws[TG_S], dw[TG_S];
#pragma unroll
        for (int s = 0; s < TG_S; ++s) {
            dls[s] = dl[s * C + c] * inv;
            ws[s] = w[s * C + c];
            dw[s] = 0.f;
        }
        for (int l = 0; l < L; ++l) {
            const float zv = zb[(size_t)l * C + c];
            float g = 0.f;
#pragma unroll
            for (int s = 0; s < TG_S; ++s) {
                g += sA[l][s] * dls[s] + sP[l][s] * ws[s];
                dw[s] += sP[l][s] * zv;
            }
            dz[((size_t)b * L + l) * C + c] = f2bf(g);
        }
#pragma unroll
        for (int s = 0; s < TG_S; ++s) dwpart[((size_t)b * TG_S + s) * C + c] = dw[s];
    }
    if (tid < TG_S) {
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += sP[l][tid];
        dbpart[(size_t)b * TG_S + tid] = t;
    }
}

int tg_tokenlearn_fwd(const float* z, const float* w, const float* bias, float* A, float* mod, int B, int L, int C, int S,
                      int G, hipStream_t st) {
    if (B <= 0 || !z || !w || !bias || !A || !mod) return SPN_ERR_ARG;
    if (S != TG_S || L > TG_MAXL || G < 0 || G + S > TG_MAXNT) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(tg_tokenlearn_fwd_kernel, dim3(B), dim3(256), 0, st, z, w, bias, A, mod, L, C, G);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

size_t tg_ws_bytes(int B, int C) { return (size_t)B * (TG_MAXNT * C + TG_MAXNT) * sizeof(float); }

int tg_tokenlearn_bwd(const float* z, const float* w, const float* A, const float* dmod, bf16_t* dz, float* dw, float* dbias,
                      float* ws, size_t ws_bytes, int B, int L, int C, int S, int G, hipStream_t st) {
    if (B <= 0 || !z || !w || !A || !dmod || !dz || !dw || !dbias || !ws) return SPN_ERR_ARG;
    if (S != TG_S || L > TG_MAXL || G < 0 || G + S > TG_MAXNT) return SPN_ERR_SHAPE;
    if (ws_bytes < tg_ws_bytes(B, C)) return SPN_ERR_WORKSPACE;
    float* dwpart = ws;
    float* dbpart = ws + (size_t)B * TG_S * C;
    hipLaunchKernelGGL(tg_tokenlearn_bwd_kernel, dim3(B), dim3(256), 0, st, z, w, A, dmod, dz, dwpart, dbpart, L, C, G);
    SPN_CHECK_LAUNCH();
    SPN_TRYG(fold_rows(dwpart, (size_t)TG_S * C, B, (size_t)TG_S * C, dw, 1.0f, 0, st));
    return fold_rows(dbpart, TG_S, B, TG_S, dbias, 1.0f, 0, st);
}

// ------------------------------------------------------------------- global tokens + GEMM operand
// mod[b, i, :] = feats[b, :] * relu(masks[i, :]) for i < G (the local tokens are already in mod);
// X[b*NT + t, :] = cat(ref[b, t, :], mod[b, t, :])   (the input of s_remain_map[0]) in bf16 (weight-gradient GEMM
// operand) and fp32 (forward: the ReLU mask is taken from an fp32 pre-activation, see tgcir_models.py)
__global__ __launch_bounds__(256) void tg_fuse_prep_kernel(const float* __restrict__ feats, const float* __restrict__ masks,
                                                          const float* __restrict__ ref, float* __restrict__ mod,
                                                          bf16_t* __restrict__ X, float* __restrict__ Xf, int C, int G,
                                                          int NT) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < NT * C; e += 256) {
        const int t = e / C, c = e % C;
        const size_t o = ((size_t)b * NT + t) * C + c;
        float m;
        if (t < G) {
            m = feats[(size_t)b * C + c] * fmaxf(masks[t * C + c], 0.f);
            mod[o] = m;
        } else {
            m = mod[o];
        }
        const float rv = ref[o];
        if (X) {
            bf16_t* xr = X + ((size_t)b * NT + t) * 2 * C;
            xr[c] = f2bf(rv);
            xr[C + c] = f2bf(m);
        }
        if (Xf) {
            float* xf = Xf + ((size_t)b * NT + t) * 2 * C;
            xf[c] = rv;
            xf[C + c] = m;
        }
    }
}

int tg_fuse_prep(const float* feats, const float* masks, const float* ref, float* mod, bf16_t* X, float* Xf, int B, int C,
                 int S, int G, hipStream_t st) {
    if (B <= 0 || !feats || !masks || !ref || !mod || (!X && !Xf)) return SPN_ERR_ARG;
    if (G + S > TG_MAXNT) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(tg_fuse_prep_kernel, dim3(B), dim3(256), 0, st, feats, masks, ref, mod, X, Xf, C, G, G + S);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// image side (frozen): mod[b, i<G, :] = feats[b] relu(masks[i]) next to the local tokens already in mod, and
// pooled[b] = mean_t mod[b, t]   (img_embed(return_pool_and_normalized), models.py:183-196; the caller normalises)
__global__ __launch_bounds__(256) void tg_img_finish_kernel(const float* __restrict__ feats, const float* __restrict__ masks,
                                                           float* __restrict__ mod, float* __restrict__ pooled, int C, int G,
                                                           int NT) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int t = 0; t < NT; ++t) {
            const size_t o = ((size_t)b * NT + t) * C + c;
            float m;
            if (t < G) {
                m = feats[(size_t)b * C + c] * fmaxf(masks[t * C + c], 0.f);
                mod[o] = m;
            } else {
                m = mod[o];
            }
            acc += m;
        }
        pooled[(size_t)b * C + c] = acc / (float)NT;
    }
}

int tg_img_finish(const float* feats, const float* masks, float* mod, float* pooled, int B, int C, int S, int G,
                  hipStream_t st) {
    if (B <= 0 || !feats || !masks || !mod || !pooled) return SPN_ERR_ARG;
    if (G + S > TG_MAXNT) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(tg_img_finish_kernel, dim3(B), dim3(256), 0, st, feats, masks, mod, pooled, C, G, G + S);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------------------- gate
// r[t] = sigmoid(<relu(hpre[t, :]), w2> + b2);  pooled = (1/NT) sum_t r[t] ref[t] + (1 - r[t]) mod[t]
__global__ __launch_bounds__(256) void tg_gate_fwd_kernel(const float* __restrict__ hpre, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, const float* __restrict__ ref,
                                                         const float* __restrict__ mod, float* __restrict__ r,
                                                         float* __restrict__ pooled, int NT, int C) {
    __shared__ float sr[TG_MAXNT];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int t = wid; t < NT; t += 4) {
        const float* h = hpre + ((size_t)b * NT + t) * C;
        float p = 0.f;
        for (int c = lane; c < C; c += 64) p += fmaxf(h[c], 0.f) * w2[c];
        p = wave_sum(p);
        if (lane == 0) {
            const float v = sigmoidf_(p + b2[0]);
            sr[t] = v;
            r[(size_t)b * NT + t] = v;
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)NT;
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int t = 0; t < NT; ++t) {
            const size_t o = ((size_t)b * NT + t) * C + c;
            acc += sr[t] * ref[o] + (1.0f - sr[t]) * mod[o];
        }
        pooled[(size_t)b * C + c] = acc * inv;
    }
}

// dfuse = dpooled / NT (every token); dr[t] = <dfuse, ref[t] - mod[t]>; e[t] = dr r (1 - r)
// dmod[t] = (1 - r[t]) dfuse;  dh[t, c] = e[t] w2[c] [hpre > 0];  dw2 = sum_t e[t] relu(hpre[t]);  db2 = sum_t e[t]
// dh leaves in fp32, row-major and transposed (the two operand layouts of the exact GEMMs for dX and dW1), with its
// column sums (= the gradient of s_remain_map[0].bias): e[t] changes sign across tokens, the sums over tokens cancel
// heavily and a bf16 dh costs ~5 % of this layer's gradient.
__global__ __launch_bounds__(256) void tg_gate_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ ref,
                                                         const float* __restrict__ mod, const float* __restrict__ r,
                                                         const float* __restrict__ hpre, const float* __restrict__ w2,
                                                         float* __restrict__ dmod, float* __restrict__ dh,
                                                         float* __restrict__ dhT, float* __restrict__ dw2part,
                                                         float* __restrict__ db1part, float* __restrict__ db2part, int NT,
                                                         int C, int ldt) {
    __shared__ float se[TG_MAXNT], sr[TG_MAXNT];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float inv = 1.0f / (float)NT;
    for (int t = wid; t < NT; t += 4) {
        float p = 0.f;
        for (int c = lane; c < C; c += 64) {
            const size_t o = ((size_t)b * NT + t) * C + c;
            p += dpooled[(size_t)b * C + c] * inv * (ref[o] - mod[o]);
        }
        p = wave_sum(p);
        if (lane == 0) {
            const float rv = r[(size_t)b * NT + t];
            sr[t] = rv;
            se[t] = p * rv * (1.0f - rv);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const float df = dpooled[(size_t)b * C + c] * inv, wv = w2[c];
        float dw = 0.f, db = 0.f;
        for (int t = 0; t < NT; ++t) {
            const size_t o = ((size_t)b * NT + t) * C + c;
            const float h = hpre[o];
            const float d = h > 0.f ? se[t] * wv : 0.f;
            dmod[o] = (1.0f - sr[t]) * df;
            dh[o] = d;
            dhT[(size_t)c * ldt + (size_t)b * NT + t] = d;
            dw += se[t] * fmaxf(h, 0.f);
            db += d;
        }
        dw2part[(size_t)b * C + c] = dw;
        db1part[(size_t)b * C + c] = db;
    }
    if (tid == 0) {
        float t0 = 0.f;
        for (int t = 0; t < NT; ++t) t0 += se[t];
        db2part[b] = t0;
    }
}

// out[0] = sum_i x[i]  (one block; n = batch size)
__global__ __launch_bounds__(256) void tg_sum_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

int tg_gate_fwd(const float* hpre, const float* w2, const float* b2, const float* ref, const float* mod, float* r,
                float* pooled, int B, int NT, int C, hipStream_t st) {
    if (B <= 0 || !hpre || !w2 || !b2 || !ref || !mod || !r || !pooled) return SPN_ERR_ARG;
    if (NT > TG_MAXNT) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(tg_gate_fwd_kernel, dim3(B), dim3(256), 0, st, hpre, w2, b2, ref, mod, r, pooled, NT, C);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int tg_gate_bwd(const float* dpooled, const float* ref, const float* mod, const float* r, const float* hpre, const float* w2,
                float* dmod, float* dh, float* dhT, float* dw2, float* db1, float* db2, float* ws, size_t ws_bytes, int B,
                int NT, int C, hipStream_t st) {
    if (B <= 0 || !dpooled || !ref || !mod || !r || !hpre || !w2 || !dmod || !dh || !dhT || !dw2 || !db1 || !db2 || !ws)
        return SPN_ERR_ARG;
    if (NT > TG_MAXNT) return SPN_ERR_SHAPE;
    if (ws_bytes < tg_ws_bytes(B, C)) return SPN_ERR_WORKSPACE;
    float* dw2part = ws;
    float* db1part = ws + (size_t)B * C;
    float* db2part = ws + 2 * (size_t)B * C;
    hipLaunchKernelGGL(tg_gate_bwd_kernel, dim3(B), dim3(256), 0, st, dpooled, ref, mod, r, hpre, w2, dmod, dh, dhT, dw2part,
                       db1part, db2part, NT, C, B * NT);
    SPN_CHECK_LAUNCH();
    SPN_TRYG(fold_rows(dw2part, (size_t)C, B, (size_t)C, dw2, 1.0f, 0, st));
    SPN_TRYG(fold_rows(db1part, (size_t)C, B, (size_t)C, db1, 1.0f, 0, st));
    hipLaunchKernelGGL(tg_sum_kernel, dim3(1), dim3(256), 0, st, db2part, B, db2);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// --------------------------------------------------------------------- back through cat + global tokens
// dmod[b, t, :] += dX[b*NT + t, C:2C];  dfeats[b] = sum_{i<G} dmod[b, i] relu(masks[i]);
// dmasks[i] = sum_b dmod[b, i] feats[b] [masks[i] > 0]
__global__ __launch_bounds__(256) void tg_mod_bwd_kernel(const float* __restrict__ dX, float* __restrict__ dmod,
                                                        const float* __restrict__ feats, const float* __restrict__ masks,
                                                        float* __restrict__ dfeats, float* __restrict__ dmpart, int C, int G,
                                                        int NT) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float df = 0.f;
        const float f = feats[(size_t)b * C + c];
        for (int t = 0; t < NT; ++t) {
            const size_t o = ((size_t)b * NT + t) * C + c;
            const float g = dmod[o] + dX[((size_t)b * NT + t) * 2 * C + C + c];
            dmod[o] = g;
            if (t < G) {
                const float m = masks[t * C + c];
                df += g * fmaxf(m, 0.f);
                dmpart[((size_t)b * G + t) * C + c] = m > 0.f ? g * f : 0.f;
            }
        }
        dfeats[(size_t)b * C + c] = df;
    }
}

int tg_mod_bwd(const float* dX, float* dmod, const float* feats, const float* masks, float* dfeats, float* dmasks, float* ws,
               size_t ws_bytes, int B, int C, int S, int G, hipStream_t st) {
    if (B <= 0 || !dX || !dmod || !feats || !masks || !dfeats || !dmasks || !ws) return SPN_ERR_ARG;
    if (G + S > TG_MAXNT || G <= 0) return SPN_ERR_SHAPE;
    if (ws_bytes < tg_ws_bytes(B, C)) return SPN_ERR_WORKSPACE;
    hipLaunchKernelGGL(tg_mod_bwd_kernel, dim3(B), dim3(256), 0, st, dX, dmod, feats, masks, dfeats, ws, C, G, G + S);
    SPN_CHECK_LAUNCH();
    return fold_rows(ws, (size_t)G * C, B, (size_t)G * C, dmasks, 1.0f, 0, st);
}

}  // namespace spn
