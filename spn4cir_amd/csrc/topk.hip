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

// One block per query row.  Order: score descending, ties by ascending index (validate.py's argsort on distinct scores;
// a stable rule for equal ones).  Round 2 ran K selection passes over the row (50 for Recall@50); this is a radix select
// on the 96-bit key (order-preserving image of the score, ~index): 8-bit digits from the top, one histogram pass per digit
// over the elements still matching the prefix - 8 passes for the score, up to 4 more only when the K-th score is tied -
// then one gathering pass and a rank sort of the K winners in LDS.  Read-only on the scores.
static constexpr int TOPK_MAX = 256;

__device__ __forceinline__ uint64_t score_key(double s) {
    const uint64_t b = (uint64_t)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);          // larger score -> larger key; -0.0 < +0.0 as bit patterns,
}                                                                  // harmless for cosines

__global__ __launch_bounds__(256) void topk_select_kernel(const double* __restrict__ scores, int Ng, int K,
                                                          const int32_t* __restrict__ exclude, int32_t* __restrict__ idx,
                                                          double* __restrict__ val) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_digit, sh_need, sh_n;
    __shared__ uint64_t wkey[TOPK_MAX];
    __shared__ int32_t widx[TOPK_MAX];
    const int row = blockIdx.x, tid = threadIdx.x;
    const double* s = scores + (size_t)row * Ng;
    const int ex = exclude ? exclude[row] : -1;
    const int cand = Ng - ((ex >= 0 && ex < Ng) ? 1 : 0);
    const int Kc = min(K, cand);                           // winners that exist
    // ---- radix select: after the loop (pk, pidx) is the composite key of the Kc-th element in the order
    uint64_t pk = 0;                                        // score-key prefix (top `round` bytes decided)
    uint32_t pi = 0;                                        // ~index prefix, rounds 8..11
    uint32_t need = (uint32_t)Kc;                           // rank still looked for inside the prefix
    bool tie_rounds = true;
    for (int round = 0; round < 12 && Kc > 0; ++round) {
        const bool on_idx = round >= 8;
        if (on_idx && !tie_rounds) break;
        hist[tid] = 0;
        __syncthreads();
        const int shift = on_idx ? 8 * (11 - round) : 8 * (7 - round);
        for (int j = tid; j < Ng; j += 256) {
            if (j == ex) continue;
            const uint64_t k = score_key(s[j]);
            const uint32_t ni = ~(uint32_t)j;
            bool match;
            uint32_t digit;
            if (!on_idx) {
                match = round == 0 || (k >> (shift + 8)) == (pk >> (shift + 8));
                digit = (uint32_t)(k >> shift) & 255u;
            } else {
                match = k == pk && (round == 8 || (ni >> (shift + 8)) == (pi >> (shift + 8)));
                digit = (ni >> shift) & 255u;
            }
            if (match) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        if (tid == 0) {                                     // largest digit first: the bin holding rank `need`
            uint32_t acc = 0;
            int d = 255;
            for (; d > 0; --d) {
                if (acc + hist[d] >= need) break;
                acc += hist[d];
            }
            sh_digit = (uint32_t)d;
            sh_need = need - acc;
            sh_n = hist[d];
        }
        __syncthreads();
        need = sh_need;
        if (!on_idx) pk |= (uint64_t)sh_digit << shift;
        else pi |= sh_digit << shift;
        // after the last score byte: `sh_n` elements carry the K-th score; if all of them are wanted the index rounds
        // have nothing to decide (the usual case: distinct scores, sh_n = need = 1)
        if (round == 7) {
            tie_rounds = sh_n != need;
            if (!tie_rounds) pi = 0;                        // every element with this score is selected
        }
        __syncthreads();
    }
    // ---- gather the winners: composite key >= (pk, pi)
    if (tid == 0) sh_n = 0;
    __syncthreads();
    if (Kc > 0) {
        for (int j = tid; j < Ng; j += 256) {
            if (j == ex) continue;
            const uint64_t k = score_key(s[j]);
            const uint32_t ni = ~(uint32_t)j;
            if (k > pk || (k == pk && ni >= pi)) {
                const uint32_t slot = atomicAdd(&sh_n, 1u);
                if (slot < (uint32_t)TOPK_MAX) { wkey[slot] = k; widx[slot] = j; }
            }
        }
    }
    __syncthreads();
    // ---- rank sort (Kc <= 256 distinct composite keys) and output; rows with fewer than K candidates end in -1
    if (tid < Kc) {
        const uint64_t k = wkey[tid];
        const int32_t j = widx[tid];
        int rank = 0;
        for (int o = 0; o < Kc; ++o) rank += (wkey[o] > k || (wkey[o] == k && widx[o] < j)) ? 1 : 0;
        idx[(size_t)row * K + rank] = j;
        if (val) val[(size_t)row * K + rank] = s[j];
    }
    for (int kk = Kc + tid; kk < K; kk += 256) {
        idx[(size_t)row * K + kk] = -1;
        if (val) val[(size_t)row * K + kk] = -INFINITY;
    }
}

int topk_from_scores(const double* scores, int Nq, int Ng, int K, const int32_t* exclude, int32_t* idx, double* val,
                     hipStream_t st) {
    if (Nq <= 0 || Ng <= 0 || K <= 0) return SPN_ERR_ARG;
    if (K > TOPK_MAX) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(topk_select_kernel, dim3(Nq), dim3(256), 0, st, scores, Ng, K, exclude, idx, val);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
