// Bank InfoNCE, forward statistics pass for 128..256 queries per call (clip4cir/models_negplus.py:150-154: logits = q bank^T /
// tau, row softmax statistics; BASELINE config 2's single-GPU shape B = 256, M = 40 000, D = 768).
//
// Why its own kernel.  The 256 x 256-tile GEMM with a statistics epilogue (gemm_nt2<.., GEMM_BANKSTATS>) is 157 workgroups at
// 40 000 bank rows, each pulling 393 KB of bank from HBM with one 32 KB k tile or so in flight: a CU draws ~20 GB/s that way, 157 of
// them ~3 TB/s, and the 61 MB bank takes >= 20 us whatever the matrix pipe does (43-45 us measured: 0.17 of the HBM roofline).
// HBM-bound work wants (a) every CU pulling and (b) enough bytes in flight per CU:
//   * tile = 160 bank rows x all (<= 256) queries: 250 workgroups at 40 000 rows, one per CU, one round;
//   * the bank operand runs through a FOUR-stage LDS ring (3 x 20 KB in flight per CU), the queries (L2-resident, 393 KB) through a
//     two-stage ring.  vmcnt is per wave and completes in order, so a wave that waits for a query k tile would also drain every
//     bank request issued before it: the two streams are issued by DIFFERENT waves - waves 0-3 own the bank DMA (5 instructions of
//     8 rows x 128 B per k tile each), waves 4-7 the query DMA (8 each) - and each counts only its own stream;
//   * wave w owns queries [32 w, 32 w + 32) against all 160 rows (5 accumulator tiles of v_mfma_f32_32x32x16_bf16, operand order
//     (bank, query): a LANE owns one query, its registers the bank rows) - the row statistics of a query are an in-lane reduction
//     over 80 registers plus one exchange with lane ^ 32, no cross-wave traffic.
// Optional save for the GEMM-shaped backward pass: p = exp(z - tile max) as bf16, written TRANSPOSED ([bank row][query] = the G^T
// operand layout of the dq GEMM: whole 512-byte rows through an LDS image), and the tile maxima; bank_gt_scale_kernel then turns it
// into G^T (row-major to row-major: no transpose).
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static constexpr int S160_ROWS = 160, S160_NB = 4, S160_NQ = 2;
static constexpr int S160_BANK_STAGE = S160_ROWS * 128, S160_Q_STAGE = 256 * 128;
static constexpr int S160_LDS = S160_NB * S160_BANK_STAGE + S160_NQ * S160_Q_STAGE;      // 147 456 B

__device__ __forceinline__ int s160_swz(int r, int c) { return c ^ ((r >> 1) & 7); }
__device__ __forceinline__ bf16x8 s160_frag(const char* sT, int r, int c) {
    return *(const bf16x8*)(sT + r * 128 + (s160_swz(r, c) << 4));
}

int bank_stats160_tiles(int M) { return (M + S160_ROWS - 1) / S160_ROWS; }

template <bool SAVE>
__global__ __launch_bounds__(512, 2) void bank_stats160_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ bank,
                                                              const int64_t* __restrict__ labels, int B, int M, int D, int m_begin,
                                                              float inv_tau, float* __restrict__ partial, bf16_t* __restrict__ Pt,
                                                              float* __restrict__ tmax, int* __restrict__ tail_counter) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // arrival counter of bank_stats_tail_kernel (bank_step at 192..256 queries): reset by this launch, in front of it
    if (tail_counter && blockIdx.x == 0 && threadIdx.x == 0) *tail_counter = 0;
    char* sBank = smem;
    char* sQ = smem + S160_NB * S160_BANK_STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, m0 = tile * S160_ROWS;
    const int nk = D / 64;
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(bank, (uint32_t)M * (uint32_t)D * 2u);
    const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(q, (uint32_t)B * (uint32_t)ldq * 2u);
    // DMA: instruction i of a k tile = 8 rows x 128 B (lane -> row i*8 + lane/8, 16-byte chunk lane%8, swizzled on the source side)
    auto dma_bank = [&](int kt) {                 // waves 0-3: 5 instructions each (20 = 160 rows / 8)
        char* sb = sBank + (kt % S160_NB) * S160_BANK_STAGE;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int i = wid * 5 + t, r = i * 8 + (lane >> 3), c = s160_swz(r, lane & 7);
            glds16(rsB, sb + i * 1024, ((uint32_t)(m0 + r) * (uint32_t)D + (uint32_t)(kt * 64 + c * 8)) * 2u);
        }
    };
    auto dma_q = [&](int kt) {                    // waves 4-7: 8 instructions each (32 = 256 rows / 8)
        char* sq = sQ + (kt % S160_NQ) * S160_Q_STAGE;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = (wid - 4) * 8 + t, r = i * 8 + (lane >> 3), c = s160_swz(r, lane & 7);
            glds16(rsQ, sq + i * 1024, ((uint32_t)r * (uint32_t)ldq + (uint32_t)(kt * 64 + c * 8)) * 2u);
        }
    };
    const bool bank_wave = wid < 4;
    if (bank_wave) {
        dma_bank(0);
        if (nk > 1) dma_bank(1);
        if (nk > 2) dma_bank(2);
    } else {
        dma_q(0);
    }
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int qrow = wid * 32 + (lane & 31), cl = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's own stream: k tile kt has landed when at most the newer requests are outstanding
        if (bank_wave) {
            const int newer = min(nk - 1 - kt, 2);                 // bank k tiles kt+1, kt+2 may stay in flight
            if (newer == 2) wait_vmcnt<10>();
            else if (newer == 1) wait_vmcnt<5>();
            else wait_vmcnt<0>();
        } else {
            wait_vmcnt<0>();                                       // the query k tile requested one iteration ago
        }
        __builtin_amdgcn_s_barrier();                              // everyone's part of k tile kt is in LDS; k tile kt-1 is consumed
        if (bank_wave) {
            if (kt + 3 < nk) dma_bank(kt + 3);                     // into the stage of k tile kt-1
        } else {
            if (kt + 1 < nk) dma_q(kt + 1);
        }
        const char* sb = sBank + (kt % S160_NB) * S160_BANK_STAGE;
        const char* sq = sQ + (kt % S160_NQ) * S160_Q_STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 qf = s160_frag(sq, qrow, kk * 2 + cl);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s160_frag(sb, i * 32 + (lane & 31), kk * 2 + cl), qf, acc[i], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // fragment reads done before the next barrier lets DMA overwrite
    }
    // ---- statistics of query n = wid * 32 + (lane & 31) over the tile's rows m = m0 + 32 i + 8 g + 4 (lane >> 5) + r
    const int n = qrow;
    const bool q_ok = n < B;
    const int64_t lab = q_ok ? labels[n] - (int64_t)m_begin : -1;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + i * 32 + (e >> 2) * 8 + cl * 4 + (e & 3);
            const float z = m < M ? acc[i][e] * inv_tau : -INFINITY;
            acc[i][e] = z;
            mx = fmaxf(mx, z);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f, sl = 0.f, lv = -INFINITY;
    const float mu = mx == -INFINITY ? 0.f : mx;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + i * 32 + (e >> 2) * 8 + cl * 4 + (e & 3);
            const float z = acc[i][e];
            const float p = __expf(z - mu);                        // exp(-inf) = 0 for rows beyond the bank
            l += p;
            sl += m < M ? z : 0.f;
            lv = ((int64_t)m == lab) ? z : lv;
            acc[i][e] = p;
        }
    l += __shfl_xor(l, 32, 64);
    sl += __shfl_xor(sl, 32, 64);
    lv = fmaxf(lv, __shfl_xor(lv, 32, 64));
    if (q_ok && cl == 0) {
        *(f32x4*)(partial + ((size_t)tile * B + n) * 4) = f32x4{mx, l, sl, lv};
        if (SAVE) tmax[(size_t)tile * B + n] = mx;
    }
    if constexpr (SAVE) {
        // P^T[m][n]: the wave's [160 x 32] block into an LDS image [160][256] bf16 (the bank ring is free), then whole rows out
        __syncthreads();
        bf16_t* sP = (bf16_t*)smem;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = i * 32 + (e >> 2) * 8 + cl * 4 + (e & 3);
                sP[r * 256 + n] = f2bf(acc[i][e]);
            }
        __syncthreads();
        for (int idx = tid; idx < S160_ROWS * 32; idx += 512) {     // 16-byte pieces: row idx / 32, queries (idx % 32) * 8 ..
            const int r = idx >> 5, c8 = (idx & 31) * 8;
            if (m0 + r < M && c8 < B) *(bf16x8*)(Pt + (size_t)(m0 + r) * B + c8) = *(const bf16x8*)(sP + r * 256 + c8);
        }
    }
}

// G^T from the saved P^T: G[m][b] = p exp(tile max - lse_b) - onehot - eps / M, bf16 [M][B] (both row-major by bank row: no
// transpose; the saved p stays intact, so a second backward call on the same buffer - another label smoothing - is valid).
// Block = 8 bank rows x (B / 8) column groups per step, walking rows blockIdx.x * 8, + gridDim.x * 8, ...
__global__ __launch_bounds__(256) void bank_gt_scale_kernel(const bf16_t* __restrict__ Pt, bf16_t* __restrict__ Gt, const float* __restrict__ tmax,
                                                           const float* __restrict__ lse, const int64_t* __restrict__ labels, int B, int M,
                                                           int m_begin, float ls, float inv_m) {
    const int groups = B >> 3;                              // 16-byte column groups per row (<= 32)
    const int rl = threadIdx.x / groups, cg = threadIdx.x % groups, rows_per = 256 / groups;
    if (rl >= rows_per) return;
    const int b0 = cg * 8;
    float ls8[8];
    int lab8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ls8[e] = lse[b0 + e];
        lab8[e] = (int)(labels[b0 + e] - (int64_t)m_begin);
    }
    for (int m = blockIdx.x * rows_per + rl; m < M; m += gridDim.x * rows_per) {
        const int tile = m / S160_ROWS;
        const size_t o = (size_t)m * B + b0;
        bf16x8 p = *(const bf16x8*)(Pt + o);
        const f32x4 t0 = *(const f32x4*)(tmax + (size_t)tile * B + b0), t1 = *(const f32x4*)(tmax + (size_t)tile * B + b0 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g = bf2f(p[e]) * __expf((e < 4 ? t0[e] : t1[e - 4]) - ls8[e]) - ls * inv_m;
            g -= (m == lab8[e]) ? 1.0f - ls : 0.f;
            p[e] = f2bf(g);
        }
        *(bf16x8*)(Gt + o) = p;
    }
}

// Fold + finalize of a single-shard step in ONE launch (bank_step at 192..256 queries): wave = one query, the arithmetic of
// bank_stats_fold_kernel (lanes stride 64 over the tiles, the same merge order) followed by bank_loss_finalize_kernel's
// single-shard row (lse = m + log l, loss = lse - label logit), so that lse - hence G and dq - is BIT-identical to the three-call
// path.  The mean is summed by the block that arrives last (ticket; release / acquire at agent scope) with the finalize kernel's
// own 256-leaf tree: bit-identical too.
__global__ __launch_bounds__(256) void bank_stats_tail_kernel(const float* __restrict__ ws, int n, int B, float* __restrict__ row_lse,
                                                             float* row_loss, float* __restrict__ loss_mean, int* counter) {
    __shared__ float red[256];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = blockIdx.x * 4 + (tid >> 6);
    if (q < B) {
        float m = -INFINITY, l = 0.f, lab = -INFINITY;
        for (int i = lane; i < n; i += 64) {
            const f32x4 p = *(const f32x4*)(ws + ((size_t)i * B + q) * 4);
            const float mn = fmaxf(m, p[0]);
            if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
            m = mn;
            lab = fmaxf(lab, p[3]);
        }
        const float mw = wave_max(m);
        l = (m > -INFINITY) ? l * __expf(m - mw) : 0.f;
        l = wave_sum(l);
        lab = wave_max(lab);
        if (lane == 0) {
            // the finalize kernel re-merges the ONE folded shard: l * exp(m - m) + ... from (m, l) = (-inf, 0): l * 1 exactly
            const float lse = mw + logf(l);
            row_lse[q] = lse;
            row_loss[q] = lse - lab;
        }
    }
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    float acc = 0.f;
    for (int i = tid; i < B; i += 256) acc += __hip_atomic_load(row_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) *loss_mean = red[0] / (float)B;
}

int bank_stats_tail(const float* partial, int ntiles, int B, float* row_lse, float* row_loss, float* loss_mean, int* counter,
                    hipStream_t st) {
    hipLaunchKernelGGL(bank_stats_tail_kernel, dim3((B + 3) / 4), dim3(256), 0, st, partial, ntiles, B, row_lse, row_loss, loss_mean,
                       counter);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

bool bank_stats160_ok(int B, int D, int ldq) { return B >= 128 && B <= 256 && B % 8 == 0 && D % 64 == 0 && D >= 192 && ldq % 8 == 0; }

int bank_stats160(const bf16_t* q, int ldq, const bf16_t* bank, const int64_t* labels, int B, int M, int D, int m_begin, float inv_tau,
                  float* partial, bf16_t* Pt, float* tmax, hipStream_t st, int* tail_counter) {
    if (!bank_stats160_ok(B, D, ldq)) return SPN_ERR_SHAPE;
    if ((uint64_t)M * D * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)bank_stats160_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S160_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bank_stats160_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S160_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = bank_stats160_tiles(M);
    if (Pt)
        hipLaunchKernelGGL(bank_stats160_kernel<true>, dim3(tiles), dim3(512), S160_LDS, st, q, ldq, bank, labels, B, M, D, m_begin, inv_tau,
                           partial, Pt, tmax, tail_counter);
    else
        hipLaunchKernelGGL(bank_stats160_kernel<false>, dim3(tiles), dim3(512), S160_LDS, st, q, ldq, bank, labels, B, M, D, m_begin, inv_tau,
                           partial, (bf16_t*)nullptr, (float*)nullptr, tail_counter);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int bank_gt_scale(const bf16_t* Pt, bf16_t* Gt, const float* tmax, const float* lse, const int64_t* labels, int B, int M, int m_begin,
                  float ls, float inv_m, hipStream_t st) {
    const int rows_per = 256 / (B >> 3);
    int blocks = (M + rows_per - 1) / rows_per;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bank_gt_scale_kernel, dim3(blocks), dim3(256), 0, st, Pt, Gt, tmax, lse, labels, B, M, m_begin, ls, inv_m);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
