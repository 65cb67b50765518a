// Recall@K support: exact cosine scores and top-K selection (validate.py:28-33,123-128).
//
// Scores are accumulated in fp64 from the fp32 inputs, so the ranking is the ranking of the
// exact products of the inputs and does not depend on summation order (SURVEY.md section 7
// hard part (g): low-precision scoring breaks top-K set identity).  Ties: lower index first.
#include "common.h"
#include "kernels.h"

namespace spn {

// out[i][j] = sum_d q[i][d] * g[j][d]   (fp64 accumulate), 32x32 output tile per block
__global__ __launch_bounds__(256) void scores_f64_kernel(const float* __restrict__ q, const float* __restrict__ g,
                                                         int Nq, int Ng, int D, double* __restrict__ out) {
    __shared__ float sq[32][33], sg[32][33];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, 2x2 outputs each
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    double acc[2][2] = {{0, 0}, {0, 0}};
    for (int d0 = 0; d0 < D; d0 += 32) {
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int r = e >> 5, c = e & 31;
            sq[r][c] = (i0 + r < Nq && d0 + c < D) ? q[(size_t)(i0 + r) * D + d0 + c] : 0.f;
            sg[r][c] = (j0 + r < Ng && d0 + c < D) ? g[(size_t)(j0 + r) * D + d0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const double a0 = sq[ty][c], a1 = sq[ty + 16][c];
            const double b0 = sg[tx][c], b1 = sg[tx + 16][c];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1;
            acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * v;
            if (i < Nq && j < Ng) out[(size_t)i * Ng + j] = acc[u][v];
        }
}

int cosine_scores_f64(const float* q, const float* gallery, int Nq, int Ng, int D, double* out, hipStream_t st) {
    if (Nq <= 0 || Ng <= 0 || D <= 0) return SPN_ERR_ARG;
    hipLaunchKernelGGL(scores_f64_kernel, dim3((Ng + 31) / 32, (Nq + 31) / 32), dim3(256), 0, st, q, gallery, Nq, Ng, D,
                       out);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// "a precedes b": higher score first, ties by lower index
__device__ __forceinline__ bool precedes(double sa, int ia, double sb, int ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// One block per query row: K selection passes, each finds the best element strictly after the
// previous winner in the (score desc, index asc) order.  Read-only on the scores.
__global__ __launch_bounds__(256) void topk_select_kernel(const double* __restrict__ scores, int Ng, int K,
                                                          const int32_t* __restrict__ exclude, int32_t* __restrict__ idx,
                                                          double* __restrict__ val) {
    __shared__ double ws[4];
    __shared__ int wi[4];
    const int row = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const double* s = scores + (size_t)row * Ng;
    const int ex = exclude ? exclude[row] : -1;
    double ps = INFINITY;
    int pi = -1;
    for (int k = 0; k < K; ++k) {
        double bs = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = threadIdx.x; j < Ng; j += 256) {
            if (j == ex) continue;
            const double v = s[j];
            if (!(pi < 0 || precedes(ps, pi, v, j))) continue;   // not after the previous winner
            if (bi == 0x7fffffff || precedes(v, j, bs, bi)) { bs = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double os = __shfl_xor(bs, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || precedes(os, oi, bs, bi))) { bs = os; bi = oi; }
        }
        if (lane == 0) { ws[wid] = bs; wi[wid] = bi; }
        __syncthreads();
        bs = ws[0]; bi = wi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wi[w] != 0x7fffffff && (bi == 0x7fffffff || precedes(ws[w], wi[w], bs, bi))) { bs = ws[w]; bi = wi[w]; }
        __syncthreads();
        if (threadIdx.x == 0) {
            idx[(size_t)row * K + k] = bi == 0x7fffffff ? -1 : bi;
            if (val) val[(size_t)row * K + k] = bs;
        }
        ps = bs; pi = bi;
        if (bi == 0x7fffffff) {   // fewer than K candidates: fill the rest with -1
            for (int kk = k + 1 + threadIdx.x; kk < K; kk += 256) {
                idx[(size_t)row * K + kk] = -1;
                if (val) val[(size_t)row * K + kk] = -INFINITY;
            }
            break;
        }
    }
}

int topk_from_scores(const double* scores, int Nq, int Ng, int K, const int32_t* exclude, int32_t* idx, double* val,
                     hipStream_t st) {
    if (Nq <= 0 || Ng <= 0 || K <= 0) return SPN_ERR_ARG;
    hipLaunchKernelGGL(topk_select_kernel, dim3(Nq), dim3(256), 0, st, scores, Ng, K, exclude, idx, val);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
