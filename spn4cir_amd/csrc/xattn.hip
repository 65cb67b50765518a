// Cross-attention of the BLIP fusion encoder with the K/V projections ABSORBED into the query and the output side
// (blip4cir/med.py:97-181 BertSelfAttention with is_cross_attention; the keys/values are projections of the SAME frozen image
// tokens X_b [S, E] for all H heads):
//
//     scores_h = (q_h Wk_h) X^T                   instead of   q_h (X Wk_h^T)^T        (the key bias adds a constant per query:
//     ctx_h    = (softmax(scores_h) X) Wv_h^T + bv_h      of   softmax(.) (X Wv_h^T + bv_h)   softmax-invariant, dropped)
//
// so a layer never forms K and V [B*S, 2W] (73 856 x 1 536 per layer in config 4: 174 GFLOP forward, 174 GFLOP for their weight
// gradient, 227 MB written and re-read twice) but works on the "absorbed" queries Q' = 0.125 q_h Wk_h [B, L*H, E]:
//     forward   Q' (K = 64)  ->  P = softmax(Q' X^T)  ->  O' = P X  ->  ctx_h = O'_h Wv_h^T + bv_h
//     backward  dO' = dctx_h Wv_h  ->  dS = P o (dO' X^T - delta)  ->  dQ' = dS X  ->  dq_h = 0.125 dQ'_h Wk_h^T,
//               dWk_h = 0.125 q_h^T dQ'_h,  dWv_h = dctx_h^T O'_h,  dbv = colsum(dctx),  dbk = 0
// = 204 instead of 369 GFLOP per layer at B = 128, L = 32, S = 577, and no stream larger than [T, H, E] bf16 (75 MB).
// All kernels are built from the 128 x 128 x 64 tile primitives of gemm_v1_tiles.h; the per-sample operands are addressed through
// buffer descriptors over exactly one sample, so the ragged S = 577 (rows 577..639 of a tile) reads as zero.
// Rows of a sample are r = l * H + h (the memory order of [T, H, E]); R = L * H; SP = S rounded up to 128.
#include <math.h>
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "gemm_v1_tiles.h"

namespace spn {

// Store a workgroup's 128 x 128 fp32 accumulator tile (4 waves, 2 x 2, each 4 x 4 MFMA tiles: row = lane & 15, 4 consecutive
// columns at (lane >> 4) * 4) as bf16 rows of 256 contiguous bytes: through an LDS image with a 272-byte row pitch (8-byte
// fragments land conflict-free), read back 16 bytes per lane - one instruction stores four whole 256-byte rows instead of sixteen
// 32-byte pieces.  `stage` >= 128 * 272 bytes, free to overwrite (the caller has synchronised after its last operand read).
static constexpr int OUT_PITCH = 272;
__device__ __forceinline__ void store_tile_bf16(const f32x4 (&acc)[4][4], float alpha, char* stage, bf16_t* out, size_t ld,
                                                int rows_valid, int wr, int wc, int lane, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        char* row = stage + (wr * 64 + i * 16 + (lane & 15)) * OUT_PITCH + (wc * 64 + (lane >> 4) * 4) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v = acc[i][j] * alpha;
            *(bf16x4*)(row + j * 32) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 16 + (tid >> 4), c = tid & 15;              // 16 rows of 16 x 16 B per pass
        if (r < rows_valid) *(u32x4*)(out + (size_t)r * ld + c * 8) = *(const u32x4*)(stage + r * OUT_PITCH + c * 16);
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// head_expand: out[t, h, :] = alpha * A[t, h*64 .. +64] . Wt[:, col0 + h*64 .. +64]^T     (K = 64: one k tile, store-bound)
// A [T, lda] bf16; Wt [E, ldw] = the TRANSPOSED bf16 copy of the K/V weight; out [T, H, E] bf16.
// ----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2) void xattn_head_expand_kernel(const bf16_t* __restrict__ A, int lda,
                                                                        const bf16_t* __restrict__ Wt, int ldw, int col0,
                                                                        bf16_t* __restrict__ out, int T, int H, int E, float alpha) {
    __shared__ __attribute__((aligned(16))) char smem[128 * OUT_PITCH];         // two operand tiles (32 KB), then the output image
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = E / BN, tiles_m = (T + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % tiles_n) * BN, m0 = ((bid / tiles_n) % tiles_m) * BM, h = bid / (tiles_n * tiles_m);
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + h * 64, ((uint32_t)T * (uint32_t)lda - (uint32_t)(h * 64)) * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(Wt + col0 + h * 64, ((uint32_t)E * (uint32_t)ldw - (uint32_t)(col0 + h * 64)) * 2u);
    nt_stage<4>(rsA, smem, m0, lda, 0, wid, lane);
    nt_stage<4>(rsB, smem + TILE_BYTES, n0, ldw, 0, wid, lane);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wait_vm0();
    __syncthreads();
    const char* sA = smem;
    const char* sB = smem + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[4], b[4];
        const int c = ks * 4 + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = nt_frag(sA, wr * 64 + i * 16 + (lane & 15), c);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = nt_frag(sB, wc * 64 + i * 16 + (lane & 15), c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
    }
    __syncthreads();                                                            // everyone has read its operand fragments
    store_tile_bf16(acc, alpha, smem, out + ((size_t)m0 * H + h) * E + n0, (size_t)H * E, T - m0, wr, wc, lane, tid);
}

// ----------------------------------------------------------------------------------------------------------------------------
// head_contract: out[t, h*64 + d] = alpha * sum_e A[t, h, e] * Wr[row0 + h*64 + d, e]  (+ bias[row0 + h*64 + d])
// A [T, H, E] bf16; Wr = the bf16 K/V weight [2W, E]; out [T, ldo] bf16.  Tile 128 rows x 64 columns (one head), waves 4 x 1,
// three LDS stages of 24 KB with counted waits (the 64-row variant of gemm_nt_kernel).
// ----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2) void xattn_head_contract_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wr,
                                                                          int row0, const float* __restrict__ bias,
                                                                          bf16_t* __restrict__ out, int ldo, int T, int H, int E,
                                                                          float alpha) {
    constexpr int A_BYTES = TILE_BYTES, STAGE = TILE_BYTES + TILE_BYTES / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 3 x STAGE
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int h = bid % H, m0 = (bid / H) * BM;
    const int lda = H * E;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (size_t)h * E, ((uint32_t)T * (uint32_t)lda - (uint32_t)(h * E)) * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(Wr + (size_t)(row0 + h * 64) * E, 64u * (uint32_t)E * 2u);
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = E / BK;
    auto stage = [&](int kt, int buf) {
        char* dst = smem + buf * STAGE;
        nt_stage<4>(rsA, dst, m0, lda, kt * BK, wid, lane);
        nt_stage<2>(rsB, dst + A_BYTES, 0, E, kt * BK, wid, lane);
    };
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<6>();
        else wait_vmcnt<0>();
        lds_barrier();
        if (kt + 2 < nk) stage(kt + 2, buf >= 1 ? buf - 1 : 2);
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2], b[4];
            const int c = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = nt_frag(sA, wid * 32 + i * 16 + (lane & 15), c);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = nt_frag(sB, j * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = j * 16 + (lane >> 4) * 4;
        const f32x4 b4 = bias ? *(const f32x4*)(bias + row0 + h * 64 + d) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wid * 32 + i * 16 + (lane & 15);
            if (m >= T) continue;
            const f32x4 v = acc[i][j] * alpha + b4;
            *(bf16x4*)(out + (size_t)m * ldo + h * 64 + d) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// scores + softmax: P[b, r, :] = softmax_n( Q'[b, r, :] . X[b, n, :] ), n < S.  One workgroup owns 64 rows of one sample and ALL
// NT * 128 key columns: it walks the NT column tiles with one continuous 3-stage k pipeline (the 64-row gemm_nt_kernel loop, A
// re-staged per column tile from L2), keeps the NT x 32 fp32 accumulators per lane, and normalises in registers.
// P [B, R, NT*128] bf16, zero in the columns >= S (the K extent of the P X product).
// ----------------------------------------------------------------------------------------------------------------------------
template <int NT, int WR, int WC, int MIW>     // WR x WC waves; a wave owns 16 * MIW rows and 128 / WC columns of every column tile
__global__ __launch_bounds__(64 * WR * WC, 2) void xattn_scores_softmax_kernel(const bf16_t* __restrict__ Q,
                                                                               const bf16_t* __restrict__ X,
                                                                               bf16_t* __restrict__ P, int R, int S, int E,
                                                                               const int32_t* __restrict__ cu, int H) {
    constexpr int BMW = 16 * MIW * WR, NW = WR * WC, NJ = 8 / WC, WCOLS = 128 / WC;
    constexpr int A_BYTES = BMW * 128, STAGE = A_BYTES + TILE_BYTES;
    constexpr int GROUPS = BMW / 8 + 16;                                 // 8-row groups of one stage (A rows, then the 128 B rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 3 x STAGE, then WC x BMW floats
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int wr = wid / WC, wc = wid % WC;
    const int tiles_r = (R + BMW - 1) / BMW;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / tiles_r, m0 = (bid % tiles_r) * BMW;
    // packed rows: sample b owns the rows cu[b]*H .. cu[b+1]*H of Q / P (R = the longest sample, for the grid only)
    const size_t r0 = cu ? (size_t)cu[b] * H : (size_t)b * R;
    if (cu) R = (cu[b + 1] - cu[b]) * H;
    if (m0 >= R) return;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(Q + r0 * E, (uint32_t)R * (uint32_t)E * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(X + (size_t)b * S * E, (uint32_t)S * (uint32_t)E * 2u);
    f32x4 acc[NT][MIW][NJ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < MIW; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = E / BK, total = nk * NT;
    int s_t = 0, s_k = 0;                       // (column tile, k tile) of the next stage request
    auto stage_next = [&](int buf) {
        char* dst = smem + buf * STAGE;
#pragma unroll
        for (int g = wid; g < GROUPS; g += NW) {        // wave-uniform: (GROUPS - wid + NW - 1) / NW requests per wave and stage
            const bool isA = g < BMW / 8;
            const int R0 = (isA ? g : g - BMW / 8) * 8;
            const int r = R0 + (lane >> 3);
            const int c = nt_swz(r, lane & 7);
            const uint32_t off = ((uint32_t)((isA ? m0 : s_t * BN) + r) * (uint32_t)E + (uint32_t)(s_k * BK + c * 8)) * 2u;
            glds16(isA ? rsA : rsB, dst + (isA ? 0 : A_BYTES) + R0 * 128, off);
        }
        if (++s_k == nk) { s_k = 0; ++s_t; }
    };
    constexpr int REQ_LO = GROUPS / NW;                 // requests per stage of the waves with wid >= GROUPS % NW
    const bool more = wid < GROUPS % NW;                // ... the others issue one more
    stage_next(0);
    if (total > 1) stage_next(1);
    int buf = 0, s = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        for (int kt = 0; kt < nk; ++kt, ++s) {
            if (s + 1 < total) {
                if (more) wait_vmcnt<REQ_LO + 1>();
                else wait_vmcnt<REQ_LO>();
            } else {
                wait_vmcnt<0>();
            }
            lds_barrier();
            if (s + 2 < total) stage_next(buf >= 1 ? buf - 1 : 2);
            const char* sA = smem + buf * STAGE;
            const char* sB = sA + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[MIW], bb[NJ];
                const int c = ks * 4 + (lane >> 4);
#pragma unroll
                for (int i = 0; i < MIW; ++i) a[i] = nt_frag(sA, (wr * MIW + i) * 16 + (lane & 15), c);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bb[j] = nt_frag(sB, wc * WCOLS + j * 16 + (lane & 15), c);
#pragma unroll
                for (int i = 0; i < MIW; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[t][i][j] = mfma16(bb[j], a[i], acc[t][i][j]);
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
    }
    // softmax over the row: a lane holds, of row ((wr*MIW + i)*16 + lane&15), the columns t*128 + wc*WCOLS + j*16 + (lane>>4)*4 + e
    float* red = (float*)(smem + 3 * STAGE);                             // [WC][BMW rows]
    float rsum[MIW];
#pragma unroll
    for (int i = 0; i < MIW; ++i) {
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = t * BN + wc * WCOLS + j * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= S) acc[t][i][j][e] = -INFINITY;
                    mx = fmaxf(mx, acc[t][i][j][e]);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (lane < 16) red[wc * BMW + (wr * MIW + i) * 16 + lane] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MIW; ++i) {
        const int row = (wr * MIW + i) * 16 + (lane & 15);
        float mx = red[row];
#pragma unroll
        for (int w = 1; w < WC; ++w) mx = fmaxf(mx, red[w * BMW + row]);
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __expf(acc[t][i][j][e] - mx);
                    acc[t][i][j][e] = p;
                    sm += p;
                }
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        rsum[i] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MIW; ++i)
        if (lane < 16) red[wc * BMW + (wr * MIW + i) * 16 + lane] = rsum[i];
    __syncthreads();
    const int SP = NT * BN;
#pragma unroll
    for (int i = 0; i < MIW; ++i) {
        const int row = (wr * MIW + i) * 16 + (lane & 15);
        const int m = m0 + row;
        if (m >= R) continue;
        float sm = red[row];
#pragma unroll
        for (int w = 1; w < WC; ++w) sm += red[w * BMW + row];          // fixed order: every wave column gets the same sum
        const float inv = 1.0f / sm;
        bf16_t* o = P + (r0 + m) * SP + wc * WCOLS + (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 v = acc[t][i][j] * inv;
                *(bf16x4*)(o + t * BN + j * 16) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// dS[b, r, n] = P[b, r, n] * (dO'[b, r, :] . X[b, n, :] - delta[b*R + r])       128 x 128 tiles, two LDS stages
// ----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2) void xattn_dscores_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ X,
                                                                    const bf16_t* __restrict__ P, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dS, int R, int S, int E, int SP,
                                                                    const int32_t* __restrict__ cu, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x 2 x TILE_BYTES
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = SP / BN, tiles_m = (R + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % tiles_n) * BN, m0 = ((bid / tiles_n) % tiles_m) * BM, b = bid / (tiles_n * tiles_m);
    const size_t r0 = cu ? (size_t)cu[b] * H : (size_t)b * R;
    if (cu) R = (cu[b + 1] - cu[b]) * H;
    if (m0 >= R) return;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(dO + r0 * E, (uint32_t)R * (uint32_t)E * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(X + (size_t)b * S * E, (uint32_t)S * (uint32_t)E * 2u);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = E / BK;
    auto stage = [&](int kt, int buf) {
        char* dst = smem + buf * 2 * TILE_BYTES;
        nt_stage<4>(rsA, dst, m0, E, kt * BK, wid, lane);
        nt_stage<4>(rsB, dst + TILE_BYTES, n0, E, kt * BK, wid, lane);
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        wait_vm0();
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sA = smem + buf * 2 * TILE_BYTES;
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], bb[4];
            const int c = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = nt_frag(sA, wr * 64 + i * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < 4; ++i) bb[i] = nt_frag(sB, wc * 64 + i * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bb[j], a[i], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= R) continue;
        const float dl = delta[r0 + m];
        const size_t o = (r0 + m) * SP + n0 + wc * 64 + (lane >> 4) * 4;
        bf16x4 p4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p4[j] = *(const bf16x4*)(P + o + j * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = bf2f(p4[j][e]) * (acc[i][j][e] - dl);
            *(bf16x4*)(dS + o + j * 16) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// out[b, r, :] = A[b, r, 0..SP) . X[b, 0..S, :]      ("NN": A K-contiguous, X reduction-major)    O' = P X,  dQ' = dS X
// 128 x 128 tiles, k tiles of 64 rows of X (rows >= S read as zero through the sample's descriptor), two LDS stages.
// ----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2) void xattn_apply_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ X,
                                                                  bf16_t* __restrict__ out, int R, int S, int E, int SP,
                                                                  const int32_t* __restrict__ cu, int H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x 2 x TILE_BYTES
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = E / BN, tiles_m = (R + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % tiles_n) * BN, m0 = ((bid / tiles_n) % tiles_m) * BM, b = bid / (tiles_n * tiles_m);
    const size_t r0 = cu ? (size_t)cu[b] * H : (size_t)b * R;
    if (cu) R = (cu[b + 1] - cu[b]) * H;
    if (m0 >= R) return;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + r0 * SP, (uint32_t)R * (uint32_t)SP * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(X + (size_t)b * S * E, (uint32_t)S * (uint32_t)E * 2u);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (S + BK - 1) / BK;
    auto stage = [&](int kt, int buf) {
        char* dst = smem + buf * 2 * TILE_BYTES;
        nt_stage<4>(rsA, dst, m0, SP, kt * BK, wid, lane);
        tn_stage(rsB, dst + TILE_BYTES, kt * BK, E, n0, wid, lane);
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        wait_vm0();
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sA = smem + buf * 2 * TILE_BYTES;
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            TnFrag fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = tn_frag(sB, wc * 64 + j * 16, ks, lane);
            bf16x8 a[4], bb[4];
            const int c = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = nt_frag(sA, wr * 64 + i * 16 + (lane & 15), c);
            wait_lgkm<0>();
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = tn_tie(fb[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bb[j], a[i], acc[i][j]);
        }
    }
    __syncthreads();                                                            // the last k tile's fragments are read
    store_tile_bf16(acc, 1.0f, smem, out + (r0 + m0) * E + n0, (size_t)E, R - m0, wr, wc, lane, tid);
}

// ----------------------------------------------------------------------------------------------------------------------------
// delta[t, h] = sum_d dctx[t, h*64 + d] * (ctx[t, h*64 + d] - bv[h*64 + d])   = sum_n P dP of the absorbed form (rows of P sum to 1)
// ----------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xattn_delta_kernel(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                                          const float* __restrict__ bv, float* __restrict__ delta, int TH, int H) {
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 4, l = threadIdx.x & 15;       // 16 lanes per (t, h)
    float s = 0.f;
    if (g < TH) {
        const int h = g % H;
        const size_t o = (size_t)g * 64 + l * 4;
        const bf16x4 a = *(const bf16x4*)(dctx + o), c = *(const bf16x4*)(ctx + o);
        const f32x4 b4 = *(const f32x4*)(bv + h * 64 + l * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += bf2f(a[e]) * (bf2f(c[e]) - b4[e]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (g < TH && l == 0) delta[g] = s;
}

// ----------------------------------------------------------------------------------------------------------------------------
// Weight gradients of the absorbed K/V projections, all layers of a launch (grid.z) and both halves (grid.y: 0 = K, 1 = V):
//     dW[kv*W + h*64 + d, e] = alpha_kv * sum_t A_kv[t, h*64 + d] * B_kv[t, h, e]        A_K = q, B_K = dQ';  A_V = dctx, B_V = O'
//     dbias[W + h*64 + d]    = sum_t dctx[t, h*64 + d]   (V half);     dbias[h*64 + d] = 0   (K half: exactly zero, see the top)
// A workgroup owns TWO heads (128 columns of A, one reduction-major tile) x 128 columns e; the two heads read different B
// operands, so a stage holds three 16 KB tiles and a wave's B tile is selected by its row half.  Reduction over t in k tiles of
// 64 rows, two stages (96 KB).
// ----------------------------------------------------------------------------------------------------------------------------
struct XattnWgradSide {
    const bf16_t* A; size_t a_stride;      // [T, W] per layer (elements between layers)
    const bf16_t* B; size_t b_stride;      // [T, H, E] per layer
    float alpha;
};
struct XattnWgradArgs {
    XattnWgradSide side[2];
    float* dW; float* dbias; size_t g_stride;      // layer 0's [2W, E] weight / [2W] bias gradient, floats between layers
    int T, W, H, E;
};

__global__ __launch_bounds__(NTHREADS, 1) void xattn_wgrad_kernel(XattnWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 x 3 x TILE_BYTES
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = g.E / BN;
    const int kv = blockIdx.y, layer = blockIdx.z;
    const int hp = blockIdx.x / tiles_n, n0 = (blockIdx.x % tiles_n) * BN;       // heads 2*hp, 2*hp + 1
    const XattnWgradSide sd = g.side[kv];
    const bf16_t* A = sd.A + sd.a_stride * layer;
    const bf16_t* B = sd.B + sd.b_stride * layer;
    const int ldb = g.H * g.E;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)g.T * (uint32_t)g.W * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)g.T * (uint32_t)ldb * 2u);
    const bool do_colsum = kv == 1 && n0 == 0 && wc == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    f32x4 acc[4][4], accs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nk = (g.T + BK - 1) / BK;
    auto stage = [&](int kt, int buf) {
        char* dst = smem + buf * 3 * TILE_BYTES;
        tn_stage(rsA, dst, kt * BK, g.W, hp * 128, wid, lane);
        tn_stage(rsB, dst + TILE_BYTES, kt * BK, ldb, (2 * hp) * g.E + n0, wid, lane);
        tn_stage(rsB, dst + 2 * TILE_BYTES, kt * BK, ldb, (2 * hp + 1) * g.E + n0, wid, lane);
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        wait_vm0();
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sA = smem + buf * 3 * TILE_BYTES;
        const char* sB = sA + (1 + wr) * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            TnFrag fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = tn_frag(sA, wr * 64 + i * 16, ks, lane);
                fb[i] = tn_frag(sB, wc * 64 + i * 16, ks, lane);
            }
            wait_lgkm<0>();
            bf16x8 a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = tn_tie(fa[i]);
                bb[i] = tn_tie(fb[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(bb[j], a[i], acc[i][j]);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accs[i] = mfma16(ones, a[i], accs[i]);
            }
        }
    }
    float* dW = g.dW + g.g_stride * layer + (size_t)kv * g.W * g.E;
    float* db = g.dbias + g.g_stride * layer + (size_t)kv * g.W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = hp * 128 + wr * 64 + i * 16 + (lane & 15);                 // row of this half = h*64 + d
        if (n0 == 0 && wc == 0 && lane < 16) db[m] = kv == 1 ? accs[i][0] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            *(f32x4*)(dW + (size_t)m * g.E + n) = acc[i][j] * sd.alpha;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------------
bool xattn_absorb_ok(int B, int L, int H, int S, int E, int W) {
    static const bool on = [] {
        const char* e = spn_env("SPN_XATTN_ABSORB");            // 0: the K/V-projection form (attention.hip) for every shape
        return !(e && e[0] == '0');
    }();
    if (!on) return false;
    if (W != H * 64 || (H & 1) || E % 128 || S < 1 || S > 640 || L < 1) return false;
    const uint64_t R = (uint64_t)L * H, T = (uint64_t)B * L;
    return R * 640 * 2 < (1ull << 32) && R * E * 2 < (1ull << 32) && T * H * E * 2 < (1ull << 32) && (uint64_t)S * E * 2 < (1ull << 32);
}

int xattn_sp(int S) { return S <= 256 ? 256 : 640; }

template <typename K>
static int xattn_lds(K kern, int bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e == hipSuccess ? SPN_OK : (int)e;
}

int xattn_head_expand(const bf16_t* A, int lda, const bf16_t* Wt, int ldw, int col0, bf16_t* out, int T, int H, int E, float alpha,
                      hipStream_t st) {
    const int tiles = H * ((T + BM - 1) / BM) * (E / BN);
    ProfScope prof(PK_GEMM_NT, 2.0 * T * H * 64.0 * E, st);
    hipLaunchKernelGGL(xattn_head_expand_kernel, dim3(tiles), dim3(NTHREADS), 0, st, A, lda, Wt, ldw, col0, out, T, H, E, alpha);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int xattn_head_contract(const bf16_t* A, const bf16_t* Wr, int row0, const float* bias, bf16_t* out, int ldo, int T, int H, int E,
                        float alpha, hipStream_t st) {
    constexpr int LDS = 3 * (TILE_BYTES + TILE_BYTES / 2);
    static const int rc0 = xattn_lds(xattn_head_contract_kernel, LDS);
    if (rc0) return rc0;
    const int tiles = H * ((T + BM - 1) / BM);
    ProfScope prof(PK_GEMM_NT, 2.0 * T * H * 64.0 * E, st);
    hipLaunchKernelGGL(xattn_head_contract_kernel, dim3(tiles), dim3(NTHREADS), LDS, st, A, Wr, row0, bias, out, ldo, T, H, E, alpha);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

template <int NT, int WR, int WC, int MIW>
static int launch_scores_softmax(const bf16_t* Q, const bf16_t* X, bf16_t* P, int B, int R, int S, int E, hipStream_t st,
                                 const int32_t* cu, int H) {
    constexpr int BMW = 16 * MIW * WR;
    constexpr int LDS = 3 * (BMW * 128 + TILE_BYTES) + WC * BMW * 4;
    static const int rc0 = xattn_lds(xattn_scores_softmax_kernel<NT, WR, WC, MIW>, LDS);
    if (rc0) return rc0;
    const int tiles = B * ((R + BMW - 1) / BMW);
    hipLaunchKernelGGL((xattn_scores_softmax_kernel<NT, WR, WC, MIW>), dim3(tiles), dim3(64 * WR * WC), LDS, st, Q, X, P, R, S, E, cu, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int xattn_scores_softmax(const bf16_t* Q, const bf16_t* X, bf16_t* P, int B, int R, int S, int E, hipStream_t st, const int32_t* cu,
                         int H) {
    // Rows per workgroup: 64 (2 x 2 waves of 32 x 64).  Measured alternatives at config 4's shape (LABNOTES.md 5.8): 128 / 192 rows
    // (8 / 12 waves, one workgroup per CU, the sample's X streamed once for 2 / 3 x the rows) 127 / 137 us against 123; 48 rows
    // (1 x 4 waves of 48 x 32, two exact rounds of 1 024 workgroups) 130 us dense / 103 packed against 121 / 108.
    ProfScope prof(PK_ATTN_FWD, 2.0 * B * R * (double)S * E, st);
    if (xattn2_on()) return xattn2_scores_softmax(Q, X, P, B, R, S, E, st, cu, H);
    if (xattn_sp(S) == 256) return launch_scores_softmax<2, 2, 2, 2>(Q, X, P, B, R, S, E, st, cu, H);
    return launch_scores_softmax<5, 2, 2, 2>(Q, X, P, B, R, S, E, st, cu, H);
}

int xattn_dscores(const bf16_t* dO, const bf16_t* X, const bf16_t* P, const float* delta, bf16_t* dS, int B, int R, int S, int E,
                  hipStream_t st, const int32_t* cu, int H) {
    constexpr int LDS = 4 * TILE_BYTES;
    static const int rc0 = xattn_lds(xattn_dscores_kernel, LDS);
    if (rc0) return rc0;
    const int SP = xattn_sp(S);
    const int tiles = B * ((R + BM - 1) / BM) * (SP / BN);
    ProfScope prof(PK_ATTN_BWD, 2.0 * B * R * (double)S * E, st);
    if (xattn2_on()) return xattn2_dscores(dO, X, P, delta, dS, B, R, S, E, st, cu, H);
    hipLaunchKernelGGL(xattn_dscores_kernel, dim3(tiles), dim3(NTHREADS), LDS, st, dO, X, P, delta, dS, R, S, E, SP, cu, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int xattn_apply(const bf16_t* A, const bf16_t* X, bf16_t* out, int B, int R, int S, int E, hipStream_t st, const int32_t* cu, int H,
                int64_t total_rows) {
    constexpr int LDS = 4 * TILE_BYTES;
    static const int rc0 = xattn_lds(xattn_apply_kernel, LDS);
    if (rc0) return rc0;
    const int tiles = B * ((R + BM - 1) / BM) * (E / BN);
    ProfScope prof(PK_ATTN_FWD, 2.0 * B * R * (double)S * E, st);
    // 192-row tiles of the 8-wave kernel against the 128-row tiles here: ragged samples (packed rows) waste the tail of their last
    // tile, so the kernel is picked by the row slots each would run for the AVERAGE sample, weighted by the measured time per
    // slot (0.88, B = 128 dense: 65.4 against 74.6 us; profiles/r06_xattn_ab.txt)
    if (xattn2_on() && xattn2_apply_ok(E)) {
        const double avg = cu && total_rows > 0 ? (double)total_rows / B : (double)R;
        const double slots2 = ceil(avg / 192.0) * 192.0 * 0.88, slots1 = ceil(avg / 128.0) * 128.0;
        if (slots2 <= slots1) return xattn2_apply(A, X, out, B, R, S, E, st, cu, H);
    }
    hipLaunchKernelGGL(xattn_apply_kernel, dim3(tiles), dim3(NTHREADS), LDS, st, A, X, out, R, S, E, xattn_sp(S), cu, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int xattn_delta(const bf16_t* dctx, const bf16_t* ctx, const float* bv, float* delta, int T, int H, hipStream_t st) {
    const int TH = T * H;
    hipLaunchKernelGGL(xattn_delta_kernel, dim3((TH * 16 + 255) / 256), dim3(256), 0, st, dctx, ctx, bv, delta, TH, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int xattn_wgrad(const bf16_t* q, size_t q_stride, const bf16_t* dqa, size_t dqa_stride, const bf16_t* dctx, size_t dctx_stride,
                const bf16_t* oa, size_t oa_stride, float* dW, float* dbias, size_t g_stride, int layers, int T, int W, int H, int E,
                float scale, hipStream_t st) {
    constexpr int LDS = 6 * TILE_BYTES;
    static const int rc0 = xattn_lds(xattn_wgrad_kernel, LDS);
    if (rc0) return rc0;
    XattnWgradArgs g;
    g.side[0] = XattnWgradSide{q, q_stride, dqa, dqa_stride, scale};
    g.side[1] = XattnWgradSide{dctx, dctx_stride, oa, oa_stride, 1.0f};
    g.dW = dW; g.dbias = dbias; g.g_stride = g_stride;
    g.T = T; g.W = W; g.H = H; g.E = E;
    ProfScope prof(PK_GEMM_TN, 2.0 * 2.0 * layers * (double)T * W * E, st);
    hipLaunchKernelGGL(xattn_wgrad_kernel, dim3((H / 2) * (E / BN), 2, layers), dim3(NTHREADS), LDS, st, g);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
