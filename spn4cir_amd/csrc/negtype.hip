// Negative-type ablation head (clip4cir/models_negtype.py:53-134): four in-batch InfoNCE terms over B x B candidate sets,
//   target  L[i][j] = <q_i, t_j> / tau          query  L[i][j] = <t_i, q_j> / tau
//   text    L[i][j] = <n(R_i + T_j), t_i> / tau  refer  L[i][j] = <n(R_j + T_i), t_i> / tau
// (q_i = n(R_i + T_i), t_i = n(I_i), n = L2-normalise; every term is CE over j with label i; the loss is the mean of the terms
// selected by the neg_type bits 8 / 4 / 2 / 1) - forward AND the gradient w.r.t. the three feature matrices in one call, fp32
// throughout.  B x B x D is small (the reference loops over the batch in Python): one wave per candidate pair for the logits,
// one workgroup per output row for the gradients, every sum in a fixed order (no atomics).  The image / text towers around it
// are the training towers of config 1 (spn_vision_fwd_train / spn_text_fwd and their backward passes).
#include "common.h"
#include "kernels.h"

namespace spn {

static constexpr float NT_EPS = 1e-12f;     // F.normalize's clamp

// layout of the workspace (floats): t [B][D] | q [B][D] | inv_t [B] | inv_q [B] | L [4][B][B] | rowloss [4][B]
struct NegtypeWs {
    float *t, *q, *inv_t, *inv_q, *L, *rowloss;
};
static NegtypeWs negtype_ws_at(float* ws, int B, int D) {
    NegtypeWs w;
    w.t = ws; w.q = w.t + (size_t)B * D; w.inv_t = w.q + (size_t)B * D; w.inv_q = w.inv_t + B;
    w.L = w.inv_q + B; w.rowloss = w.L + (size_t)4 * B * B;
    return w;
}
size_t negtype_workspace_bytes(int B, int D) { return ((size_t)2 * B * D + 2 * B + (size_t)4 * B * B + 4 * B) * sizeof(float); }

template <int NV>
__device__ __forceinline__ void load_row(const float* p, int lane, float (&v)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = p[lane + 64 * k];
}

// t = n(I), q = n(R + T): one wave per row
template <int NV>
__global__ __launch_bounds__(64) void negtype_norm_kernel(const float* __restrict__ R, const float* __restrict__ T,
                                                         const float* __restrict__ I, int D, NegtypeWs w) {
    const int i = blockIdx.x, lane = threadIdx.x;
    float r[NV], t[NV], x[NV];
    load_row<NV>(R + (size_t)i * D, lane, r);
    load_row<NV>(T + (size_t)i * D, lane, t);
    load_row<NV>(I + (size_t)i * D, lane, x);
    float sq = 0.f, si = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { r[k] += t[k]; sq += r[k] * r[k]; si += x[k] * x[k]; }
    const float iq = 1.0f / fmaxf(sqrtf(wave_sum(sq)), NT_EPS), ii = 1.0f / fmaxf(sqrtf(wave_sum(si)), NT_EPS);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        w.q[(size_t)i * D + lane + 64 * k] = r[k] * iq;
        w.t[(size_t)i * D + lane + 64 * k] = x[k] * ii;
    }
    if (lane == 0) { w.inv_q[i] = iq; w.inv_t[i] = ii; }
}

// L[k][i][j] for the enabled terms: one wave per (i, j)
template <int NV>
__global__ __launch_bounds__(64) void negtype_logits_kernel(const float* __restrict__ R, const float* __restrict__ T, int B, int D,
                                                           float inv_tau, int neg_type, NegtypeWs w) {
    const int i = blockIdx.y, j = blockIdx.x, lane = threadIdx.x;
    float ti[NV];
    load_row<NV>(w.t + (size_t)i * D, lane, ti);
    const size_t o = (size_t)i * B + j, BB = (size_t)B * B;
    if (neg_type & 4) {
        float a[NV], b[NV], s = 0.f;
        load_row<NV>(w.q + (size_t)i * D, lane, a);
        load_row<NV>(w.t + (size_t)j * D, lane, b);
#pragma unroll
        for (int k = 0; k < NV; ++k) s += a[k] * b[k];
        s = wave_sum(s);
        if (lane == 0) w.L[0 * BB + o] = s * inv_tau;
    }
    if (neg_type & 8) {
        float b[NV], s = 0.f;
        load_row<NV>(w.q + (size_t)j * D, lane, b);
#pragma unroll
        for (int k = 0; k < NV; ++k) s += ti[k] * b[k];
        s = wave_sum(s);
        if (lane == 0) w.L[1 * BB + o] = s * inv_tau;
    }
    if (neg_type & 2) {                               // v = R_i + T_j
        float a[NV], b[NV], s = 0.f, n2 = 0.f;
        load_row<NV>(R + (size_t)i * D, lane, a);
        load_row<NV>(T + (size_t)j * D, lane, b);
#pragma unroll
        for (int k = 0; k < NV; ++k) { a[k] += b[k]; n2 += a[k] * a[k]; s += a[k] * ti[k]; }
        s = wave_sum(s); n2 = wave_sum(n2);
        if (lane == 0) w.L[2 * BB + o] = s / fmaxf(sqrtf(n2), NT_EPS) * inv_tau;
    }
    if (neg_type & 1) {                               // v = R_j + T_i
        float a[NV], b[NV], s = 0.f, n2 = 0.f;
        load_row<NV>(R + (size_t)j * D, lane, a);
        load_row<NV>(T + (size_t)i * D, lane, b);
#pragma unroll
        for (int k = 0; k < NV; ++k) { a[k] += b[k]; n2 += a[k] * a[k]; s += a[k] * ti[k]; }
        s = wave_sum(s); n2 = wave_sum(n2);
        if (lane == 0) w.L[3 * BB + o] = s / fmaxf(sqrtf(n2), NT_EPS) * inv_tau;
    }
}

// row softmax: L[k][i][:] -> G = (softmax - onehot(i)) * gscale (d loss / d logit), rowloss[k][i] = lse - L[i][i]; one wave per (k, i)
__global__ __launch_bounds__(64) void negtype_softmax_kernel(int B, int neg_type, float gscale, NegtypeWs w) {
    const int i = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
    const int bit[4] = {4, 8, 2, 1};
    if (!(neg_type & bit[k])) return;
    float* row = w.L + ((size_t)k * B + i) * B;
    float m = -INFINITY;
    for (int j = lane; j < B; j += 64) m = fmaxf(m, row[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < B; j += 64) s += __expf(row[j] - m);
    s = wave_sum(s);
    const float lse = m + logf(s), lii = row[i];
    for (int j = lane; j < B; j += 64) row[j] = (__expf(row[j] - lse) - (j == i ? 1.0f : 0.0f)) * gscale;
    if (lane == 0) w.rowloss[(size_t)k * B + i] = lse - lii;
}

__global__ __launch_bounds__(64) void negtype_loss_kernel(int B, int neg_type, int cnt, NegtypeWs w, float* __restrict__ loss) {
    const int lane = threadIdx.x;
    const int bit[4] = {4, 8, 2, 1};
    float tot = 0.f;
    for (int k = 0; k < 4; ++k) {
        if (!(neg_type & bit[k])) continue;
        float s = 0.f;
        for (int i = lane; i < B; i += 64) s += w.rowloss[(size_t)k * B + i];
        tot += wave_sum(s);
    }
    if (lane == 0) *loss = tot / (float)(B * cnt);
}

// Gradients: blockIdx.y = 0: dR row, 1: dT row, 2: dI row; blockIdx.x = the row; 4 waves split the partner index p = w, w + 4, ...
// (G already carries 1 / (B cnt); g = G * inv_tau is d loss / d <.,.>).
template <int NV>
__global__ __launch_bounds__(256) void negtype_grad_kernel(const float* __restrict__ R, const float* __restrict__ T, int B, int D,
                                                          float inv_tau, int neg_type, NegtypeWs w, float* __restrict__ dR,
                                                          float* __restrict__ dT, float* __restrict__ dI) {
    __shared__ float red[2][4][NV * 64];
    const int a = blockIdx.x, which = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t BB = (size_t)B * B;
    const float* Gt = w.L;                      // [0] target, [1] query, [2] text, [3] refer
    float acc[NV], dq[NV];                      // acc: direct d(row); dq: d q_a (rows R / T only), pushed through the normalisation below
#pragma unroll
    for (int k = 0; k < NV; ++k) { acc[k] = 0.f; dq[k] = 0.f; }
    float ra[NV], ta[NV], tt[NV];
    load_row<NV>(R + (size_t)a * D, lane, ra);
    load_row<NV>(T + (size_t)a * D, lane, ta);
    load_row<NV>(w.t + (size_t)a * D, lane, tt);
    for (int p = wv; p < B; p += 4) {
        float x[NV], y[NV];
        if (which < 2) {
            // d q_a from the target (i = a, j = p) and query (i = p, j = a) terms
            if (neg_type & 12) {
                load_row<NV>(w.t + (size_t)p * D, lane, x);
                const float g = ((neg_type & 4) ? Gt[0 * BB + (size_t)a * B + p] : 0.f) + ((neg_type & 8) ? Gt[1 * BB + (size_t)p * B + a] : 0.f);
#pragma unroll
                for (int k = 0; k < NV; ++k) dq[k] += g * inv_tau * x[k];
            }
            if (which == 0) {
                if (neg_type & 2) {             // text term, i = a, j = p: v = R_a + T_p against t_a
                    load_row<NV>(T + (size_t)p * D, lane, x);
                    float n2 = 0.f, s = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) { x[k] += ra[k]; n2 += x[k] * x[k]; s += x[k] * tt[k]; }
                    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), NT_EPS);
                    s = wave_sum(s) * inv;
                    const float g = Gt[2 * BB + (size_t)a * B + p] * inv_tau * inv;
#pragma unroll
                    for (int k = 0; k < NV; ++k) acc[k] += g * (tt[k] - x[k] * inv * s);
                }
                if (neg_type & 1) {             // refer term, i = p, j = a: v = R_a + T_p against t_p
                    load_row<NV>(T + (size_t)p * D, lane, x);
                    load_row<NV>(w.t + (size_t)p * D, lane, y);
                    float n2 = 0.f, s = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) { x[k] += ra[k]; n2 += x[k] * x[k]; s += x[k] * y[k]; }
                    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), NT_EPS);
                    s = wave_sum(s) * inv;
                    const float g = Gt[3 * BB + (size_t)p * B + a] * inv_tau * inv;
#pragma unroll
                    for (int k = 0; k < NV; ++k) acc[k] += g * (y[k] - x[k] * inv * s);
                }
            } else {
                if (neg_type & 2) {             // text term, i = p, j = a: v = R_p + T_a against t_p
                    load_row<NV>(R + (size_t)p * D, lane, x);
                    load_row<NV>(w.t + (size_t)p * D, lane, y);
                    float n2 = 0.f, s = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) { x[k] += ta[k]; n2 += x[k] * x[k]; s += x[k] * y[k]; }
                    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), NT_EPS);
                    s = wave_sum(s) * inv;
                    const float g = Gt[2 * BB + (size_t)p * B + a] * inv_tau * inv;
#pragma unroll
                    for (int k = 0; k < NV; ++k) acc[k] += g * (y[k] - x[k] * inv * s);
                }
                if (neg_type & 1) {             // refer term, i = a, j = p: v = R_p + T_a against t_a
                    load_row<NV>(R + (size_t)p * D, lane, x);
                    float n2 = 0.f, s = 0.f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) { x[k] += ta[k]; n2 += x[k] * x[k]; s += x[k] * tt[k]; }
                    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(n2)), NT_EPS);
                    s = wave_sum(s) * inv;
                    const float g = Gt[3 * BB + (size_t)a * B + p] * inv_tau * inv;
#pragma unroll
                    for (int k = 0; k < NV; ++k) acc[k] += g * (tt[k] - x[k] * inv * s);
                }
            }
        } else {
            // d t_a: target (i = p, j = a) and query (i = a, j = p): q_p; text / refer (i = a, j = p): the normalised candidate
            if (neg_type & 12) {
                load_row<NV>(w.q + (size_t)p * D, lane, x);
                const float g = ((neg_type & 4) ? Gt[0 * BB + (size_t)p * B + a] : 0.f) + ((neg_type & 8) ? Gt[1 * BB + (size_t)a * B + p] : 0.f);
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += g * inv_tau * x[k];
            }
            if (neg_type & 2) {                 // v = R_a + T_p
                load_row<NV>(T + (size_t)p * D, lane, x);
                float n2 = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) { x[k] += ra[k]; n2 += x[k] * x[k]; }
                const float g = Gt[2 * BB + (size_t)a * B + p] * inv_tau / fmaxf(sqrtf(wave_sum(n2)), NT_EPS);
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += g * x[k];
            }
            if (neg_type & 1) {                 // v = R_p + T_a
                load_row<NV>(R + (size_t)p * D, lane, x);
                float n2 = 0.f;
#pragma unroll
                for (int k = 0; k < NV; ++k) { x[k] += ta[k]; n2 += x[k] * x[k]; }
                const float g = Gt[3 * BB + (size_t)a * B + p] * inv_tau / fmaxf(sqrtf(wave_sum(n2)), NT_EPS);
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += g * x[k];
            }
        }
    }
    // the four waves' partial sums, added in wave order by wave 0
#pragma unroll
    for (int k = 0; k < NV; ++k) { red[0][wv][lane + 64 * k] = acc[k]; red[1][wv][lane + 64 * k] = dq[k]; }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        acc[k] = ((red[0][0][lane + 64 * k] + red[0][1][lane + 64 * k]) + red[0][2][lane + 64 * k]) + red[0][3][lane + 64 * k];
        dq[k] = ((red[1][0][lane + 64 * k] + red[1][1][lane + 64 * k]) + red[1][2][lane + 64 * k]) + red[1][3][lane + 64 * k];
    }
    if (which < 2) {
        // q_a = v_a / |v_a|: d v_a = (d q_a - q_a <q_a, d q_a>) / |v_a|, shared by dR_a and dT_a
        float qa[NV], s = 0.f;
        load_row<NV>(w.q + (size_t)a * D, lane, qa);
#pragma unroll
        for (int k = 0; k < NV; ++k) s += qa[k] * dq[k];
        s = wave_sum(s);
        const float iq = w.inv_q[a];
        float* out = (which == 0 ? dR : dT) + (size_t)a * D;
#pragma unroll
        for (int k = 0; k < NV; ++k) out[lane + 64 * k] = acc[k] + (dq[k] - qa[k] * s) * iq;
    } else {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += tt[k] * acc[k];
        s = wave_sum(s);
        const float ii = w.inv_t[a];
#pragma unroll
        for (int k = 0; k < NV; ++k) dI[(size_t)a * D + lane + 64 * k] = (acc[k] - tt[k] * s) * ii;
    }
}

template <int NV>
static int negtype_run(const float* R, const float* T, const float* I, int B, int D, float inv_tau, int neg_type, int cnt,
                       float* loss, float* dR, float* dT, float* dI, float* ws, hipStream_t st) {
    const NegtypeWs w = negtype_ws_at(ws, B, D);
    hipLaunchKernelGGL(negtype_norm_kernel<NV>, dim3(B), dim3(64), 0, st, R, T, I, D, w);
    hipLaunchKernelGGL(negtype_logits_kernel<NV>, dim3(B, B), dim3(64), 0, st, R, T, B, D, inv_tau, neg_type, w);
    hipLaunchKernelGGL(negtype_softmax_kernel, dim3(B, 4), dim3(64), 0, st, B, neg_type, 1.0f / (float)(B * cnt), w);
    hipLaunchKernelGGL(negtype_loss_kernel, dim3(1), dim3(64), 0, st, B, neg_type, cnt, w, loss);
    hipLaunchKernelGGL(negtype_grad_kernel<NV>, dim3(B, 3), dim3(256), 0, st, R, T, B, D, inv_tau, neg_type, w, dR, dT, dI);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int negtype_head(const float* R, const float* T, const float* I, int B, int D, float inv_tau, int neg_type, float* loss,
                 float* dR, float* dT, float* dI, float* ws, size_t ws_bytes, hipStream_t st) {
    if (!R || !T || !I || !loss || !dR || !dT || !dI || !ws || B <= 0) return SPN_ERR_ARG;
    if (neg_type < 1 || neg_type > 15) return SPN_ERR_ARG;
    if (D % 64 || D < 64 || D > 1024) return SPN_ERR_SHAPE;
    if (ws_bytes < negtype_workspace_bytes(B, D)) return SPN_ERR_WORKSPACE;
    const int cnt = __builtin_popcount((unsigned)neg_type);
#define SPN_NEGTYPE(NV_) case NV_: return negtype_run<NV_>(R, T, I, B, D, inv_tau, neg_type, cnt, loss, dR, dT, dI, ws, st);
    switch (D / 64) {
        SPN_NEGTYPE(1) SPN_NEGTYPE(2) SPN_NEGTYPE(3) SPN_NEGTYPE(4) SPN_NEGTYPE(5) SPN_NEGTYPE(6) SPN_NEGTYPE(7) SPN_NEGTYPE(8)
        SPN_NEGTYPE(10) SPN_NEGTYPE(12) SPN_NEGTYPE(16)
        default: return SPN_ERR_SHAPE;
    }
#undef SPN_NEGTYPE
}

}  // namespace spn
