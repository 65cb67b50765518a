// Scaled-negative bank InfoNCE: combiner + L2-normalise, streaming similarity + online
// softmax statistics over the static negative bank, and the gradient w.r.t. the queries.
// Restates models_negplus.py:130-154 (bank_large_step / infonce_loss) without ever writing
// the B x M logits.
//
// Structure ("attention with K = V = bank"): a block owns a 32-query tile and one contiguous
// chunk of bank rows; bank tiles of 32 rows x D (bf16) are DMA'd into LDS (buffer_load...lds,
// double buffered) and each tile is used twice from LDS:
//   logits   S[q][m]  = sum_d q[q][d] bank[m][d]      bank rows read d-contiguous (ds_read_b128)
//   gradient dq[q][d] += sum_m G[q][m] bank[m][d]     bank rows read m-contiguous (ds_read_b64_tr_b16)
// Work split inside the block: wave w owns the d-slice [w*D/4, (w+1)*D/4): its q fragments stay
// in registers for the whole kernel, it produces a partial S over its slice (summed through LDS)
// and accumulates dq for its slice.  With B_local = 32 (8-way data parallel at B = 256) the bank
// is read from HBM exactly once per pass; with several query tiles the blocks that share a bank
// chunk are co-scheduled on one XCD so the chunk is served from that XCD's L2.
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static constexpr int BQ = 32;   // queries per block
static constexpr int TR = 32;   // bank rows per LDS tile

__device__ __forceinline__ f32x4 mfma16b(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ combiner + normalise
// q = normalize(refer_bank[ref_idx] + text)   (models_negplus.py:133-137; F.normalize eps 1e-12)
// A reference row outside [0, n_refer) is not dereferenced: the whole query row (and inv_norm) becomes NaN.
__global__ void combine_l2norm_fwd_kernel(const float* __restrict__ refer, const int64_t* __restrict__ ref_idx,
                                          int64_t n_refer, const float* __restrict__ text, float* __restrict__ qf,
                                          bf16_t* __restrict__ qb, float* __restrict__ inv_norm, int B, int D, int ldq) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t ri = refer ? ref_idx[b] : 0;
    if (refer && n_refer > 0 && (ri < 0 || ri >= n_refer)) {
        const float nan = __builtin_nanf("");
        if (lane == 0 && inv_norm) inv_norm[b] = nan;
        for (int c = lane * 4; c < ldq; c += 256) {
            if (c < D && qf) *(f32x4*)(qf + (size_t)b * D + c) = f32x4{nan, nan, nan, nan};
            if (qb) {
                const bf16_t v = f2bf(c < D ? nan : 0.f);
                *(bf16x4*)(qb + (size_t)b * ldq + c) = bf16x4{v, v, v, v};
            }
        }
        return;
    }
    const float* r = refer ? refer + (size_t)ri * D : nullptr;
    const float* t = text + (size_t)b * D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 x = *(const f32x4*)(t + c);
        if (r) x += *(const f32x4*)(r + c);
        ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    if (lane == 0 && inv_norm) inv_norm[b] = inv;
    for (int c = lane * 4; c < ldq; c += 256) {
        f32x4 x = {0, 0, 0, 0};
        if (c < D) {
            x = *(const f32x4*)(t + c);
            if (r) x += *(const f32x4*)(r + c);
            x *= inv;
            if (qf) *(f32x4*)(qf + (size_t)b * D + c) = x;
        }
        if (qb) {
            bf16x4 o = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3])};
            *(bf16x4*)(qb + (size_t)b * ldq + c) = o;
        }
    }
}

int combine_l2norm_fwd(const float* refer_bank, const int64_t* ref_idx, int64_t n_refer, const float* text,
                       float* q_f32, bf16_t* q_bf16, float* inv_norm, int B, int D, int ldq, hipStream_t st) {
    if (B <= 0 || D <= 0) return SPN_ERR_ARG;
    if (D % 4 || ldq % 4 || ldq < D || (refer_bank && !ref_idx)) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(combine_l2norm_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, refer_bank, ref_idx, n_refer, text,
                       q_f32, q_bf16, inv_norm, B, D, ldq);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// x = r + t, q = x * inv  =>  dx = inv * (dq - q * <q, dq>)  (valid while ||x|| > eps); dtext = dx
// scale_dev (optional): every output is multiplied by *scale_dev - the incoming d(loss) of an autograd backward, read on
// the device so that the host never synchronises on it (the map dq -> dtext is linear)
__global__ void combine_l2norm_bwd_kernel(const float* __restrict__ q, const float* __restrict__ inv_norm,
                                          const float* __restrict__ dq, float* __restrict__ dtext, int B, int D,
                                          const float* __restrict__ scale_dev) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float dot = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(q + (size_t)b * D + c), g = *(const f32x4*)(dq + (size_t)b * D + c);
        dot += a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
    }
    dot = wave_sum(dot);
    const float inv = inv_norm[b] * (scale_dev ? *scale_dev : 1.0f);
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(q + (size_t)b * D + c), g = *(const f32x4*)(dq + (size_t)b * D + c);
        *(f32x4*)(dtext + (size_t)b * D + c) = (g - a * dot) * inv;
    }
}

int combine_l2norm_bwd(const float* q, const float* inv_norm, const float* dq, float* dtext, int B, int D,
                       hipStream_t st, const float* scale_dev) {
    if (B <= 0 || D <= 0) return SPN_ERR_ARG;
    if (D % 4) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(combine_l2norm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, q, inv_norm, dq, dtext, B, D,
                       scale_dev);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------ bank streaming
// LDS image of a bank tile [TR rows][D] bf16: row r at byte r*2D; the 16-byte chunk holding
// logical columns 8c..8c+7 sits at chunk position c ^ swz(r&15).  swz is a bit permutation
// chosen so that (a) 16 consecutive rows at one logical chunk (ds_read_b128 logit fragments)
// and (b) rows {r..r+3, r+8..r+11} at one logical 32-byte pair (ds_read_b64_tr_b16 gradient
// fragments) both spread over all 64 banks.
__device__ __forceinline__ int bank_swz(int r) {
    return (((r & 3) | (((r >> 3) & 1) << 2)) << 1) | ((r >> 2) & 1);
}

// LDS bytes of bank_stream_kernel's tile area (its RS / TILES_B)
template <int D, bool FP8>
static constexpr size_t bank_stream_tiles_bytes() {
    return FP8 ? (size_t)TR * D * 2 + (size_t)((D <= 768) ? 3 : 2) * TR * D : 2 * (size_t)TR * D * 2;
}

struct BankChunking {
    int nq;         // query tiles
    int nchunks;    // bank chunks
    int rows;       // rows per chunk (multiple of TR)
};

static BankChunking bank_chunking(int B, int M, int blocks = 256) {
    BankChunking c;
    c.nq = (B + BQ - 1) / BQ;
    int n = blocks / c.nq;
    if (n < 1) n = 1;
    const int tiles = (M + TR - 1) / TR;
    if (n > tiles) n = tiles;
    const int tiles_per = (tiles + n - 1) / n;
    c.rows = tiles_per * TR;
    c.nchunks = (M + c.rows - 1) / c.rows;
    return c;
}

// e4m3 helpers shared by the stream kernels (layout notes: "fp8 bank, forward pass on the fp8 MFMA" below)
template <int D>
__device__ __forceinline__ int fp8_swz(int r) {
    if constexpr (D % 256 == 0) return r & 15;
    else return (r >> 1) & 7;
}

__device__ __forceinline__ long pack_fp8x8(const float (&v)[8]) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

__device__ __forceinline__ void unpack_fp8x8(long p, float (&v)[8]) {
    const int lo = (int)(unsigned)((unsigned long)p & 0xffffffffu), hi = (int)(unsigned)((unsigned long)p >> 32);
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1]; v[6] = d[0]; v[7] = d[1];
}

// GRP (token-max banks, blip24cir .../blip2_qformer_cir_align_prompt.py:253-265): the bank holds TR = 32 token rows per
// target, a bank tile is exactly one target, and the logit of (query, target) is the MAX over the tile's 32 rows;
// labels, m_begin and the statistics count targets.  The gradient flows to the arg-max row only (first index on
// ties, as torch.max), so G has one non-zero per (query, tile) and the dq GEMM is unchanged.
// FUSED (with BWD): ONE pass over the bank yields both the softmax statistics and the query gradient - the gradient of the
// InfoNCE loss w.r.t. q is attention with K = V = bank, dq_b = (sum_j p_bj bank_j - bank_label) / tau, so the flash-attention
// recurrence applies: every tile's G = exp(z - m) against a running row maximum m, accumulators rescaled by exp(m_old -
// m_new) when it moves.  The block writes O = sum_j exp(z_j - m_c) bank_j (fp32, relative to its chunk's final maximum m_c)
// and {m_c, l_c, sum z, label logit}; bank_fused_combine_kernel folds the chunks with the GLOBAL lse (so shards of a
// data-parallel bank combine exactly as before) and subtracts the label row.  The bank is read once per STEP instead of once
// per pass: 61 MB instead of 122.9 MB at 40 000 x 768.
template <int D, bool BWD, bool FP8, bool GRP, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void bank_stream_kernel(BankArgs a, BankChunking ck, const float* __restrict__ row_lse,
                                                            float label_smoothing, float inv_m_total,
                                                            float* __restrict__ ws, float* __restrict__ ws2 = nullptr) {
    static_assert(!FUSED || (BWD && !GRP), "the fused pass is the backward structure of a plain bank");
    constexpr int DW = D / 4;            // columns per wave
    constexpr int KSW = DW / 32;         // 32-deep k-steps per wave in the logit GEMM
    constexpr int NDT = DW / 16;         // 16-wide d tiles per wave in the gradient GEMM
    constexpr int ROWB = D * 2;          // bytes per bank row
    constexpr int TILE_B = TR * ROWB;
    constexpr int GLDS_PER_WAVE = D / 64;
    constexpr int LDG = TR + 8;          // G row stride (elements): 80 B rows, 16-B aligned
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // e4m3 bank: RS raw tiles (1 byte per element) behind ONE bf16 image; three where they fit, so that two tiles are in
    // flight per CU while a third is consumed (with one in flight the statistics and dq phases of a tile delay the request
    // for the tile after next: ablation at 400 000 rows, 144 us = 113 streaming + 19 statistics + 13 dq, nothing hidden)
    constexpr int RS = (FP8 && D <= 768) ? 3 : 2;
    constexpr int TILES_B = FP8 ? TILE_B + RS * (TR * D) : 2 * TILE_B;
    char* tiles = smem;                                  // 2 x TILE_B (bf16) / TILE_B + RS x raw (e4m3)
    float* Sp = (float*)(smem + TILES_B);                // [4 waves][2 mt][2 nt][64 lanes][4]
    bf16_t* Gs = (bf16_t*)(smem + TILES_B + 4 * 4096);   // [BQ][LDG]
    float* Fin = (float*)(smem + TILES_B);               // reuse of Sp at the end (forward stats)
    [[maybe_unused]] float* As = (float*)(smem + TILES_B + 4 * 4096 + BQ * LDG * 2);   // FUSED: rescale factor per query
    // F8L (fused pass over an e4m3 bank): the logits run on the fp8 MFMA straight from the raw tile - queries as two e4m3
    // terms, as bank_fp8_fwd_kernel - while the SAME raw tile is dequantised into the bf16 image the dq GEMM reads
    // (its G operand needs more than e4m3's 3 mantissa bits); no barrier between the two, one fewer per tile
    constexpr bool F8L = FP8 && FUSED;
    [[maybe_unused]] float* Qm = As + BQ;                 // [4 waves][32 queries] partial max |q|
    [[maybe_unused]] float* Sst = Qm + 4 * BQ;            // [2][TR] row scales of the tile in flight / in use

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mi = blockIdx.x % ck.nchunks, qi = blockIdx.x / ck.nchunks;
    const int q0 = qi * BQ;
    const int m_lo = mi * ck.rows;
    const int m_hi = min(a.M, m_lo + ck.rows);
    const int ntiles = m_hi > m_lo ? (m_hi - m_lo + TR - 1) / TR : 0;

    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bank, (uint32_t)a.M * (uint32_t)(FP8 ? D : ROWB));

    // query fragments of this wave's d-slice (B operand: j = query, k = d)
    bf16x8 qf[2][KSW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int qr = q0 + mt * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            if (qr < a.B) {
                qf[mt][ks] = *(const bf16x8*)(a.q + (size_t)qr * a.ldq + w * DW + ks * 32 + (lane >> 4) * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[mt][ks][e] = (bf16_t)0.0f;
            }
        }
    }

    [[maybe_unused]] long qh[2][F8L ? KSW : 1], ql[2][F8L ? KSW : 1];
    [[maybe_unused]] float sq[2] = {1.f, 1.f};
    if constexpr (F8L) {        // q ~= sq hi + (sq / 16) lo, one scale per query (bank_fp8_fwd_kernel; oracle: split_query_e4m3)
        float am[2] = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) am[mt] = fmaxf(am[mt], fabsf(bf2f(qf[mt][ks][e])));
            am[mt] = fmaxf(am[mt], __shfl_xor(am[mt], 16, 64));
            am[mt] = fmaxf(am[mt], __shfl_xor(am[mt], 32, 64));
            if ((lane >> 4) == 0) Qm[w * 32 + mt * 16 + lane] = am[mt];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float m = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) m = fmaxf(m, Qm[ww * 32 + mt * 16 + (lane & 15)]);
            sq[mt] = m > 0.f ? m / 448.0f : 1.0f;
            const float rh = 1.0f / sq[mt], rl = rh * 16.0f;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                float v[8], hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(__fmul_rn(bf2f(qf[mt][ks][e]), rh), -448.0f), 448.0f);
                qh[mt][ks] = pack_fp8x8(v);
                unpack_fp8x8(qh[mt][ks], hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float res = __fsub_rn(bf2f(qf[mt][ks][e]), __fmul_rn(hv[e], sq[mt]));
                    v[e] = fminf(fmaxf(__fmul_rn(res, rl), -448.0f), 448.0f);
                }
                ql[mt][ks] = pack_fp8x8(v);
            }
        }
    }

    // softmax-phase ownership: wave w owns query sub-tile mt_o and key sub-tile nt_o of each tile;
    // lane owns query q_o and keys nt_o*16 + (lane>>4)*4 + 0..3
    const int mt_o = w >> 1, nt_o = w & 1;
    const int q_o = q0 + mt_o * 16 + (lane & 15);
    const bool q_ok = q_o < a.B;
    const int64_t label = q_ok ? a.labels[q_o] - (int64_t)a.m_begin : -1;   // GRP: shard-local TARGET id
    [[maybe_unused]] const float zs = sq[mt_o] * a.inv_tau;               // F8L: query scale x 1 / tau
    float lse = 0.f;
    if constexpr (BWD && !FUSED) lse = q_ok ? row_lse[q_o] : 0.f;
    float st_m = -INFINITY, st_l = 0.f, st_sl = 0.f, st_lab = -INFINITY;

    f32x4 dq[2][NDT];
    if constexpr (BWD) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[mt][dt] = f32x4{0, 0, 0, 0};
    }

    // bf16 bank: two swizzled bf16 tiles, filled directly by the DMA.
    // fp8 bank (FP8): ONE bf16 tile at tiles[0] + two raw e4m3 tiles behind it (same LDS total); the raw tile
    // is a linear copy of TR contiguous rows, dequantised (x row scale) into the swizzled bf16 image by all
    // 256 threads, so everything downstream is unchanged and HBM sees one byte per bank element.
    constexpr int RAW_B = TR * D;
    auto stage = [&](int t, int buf) {
        const int mrow0 = m_lo + t * TR;
        if constexpr (FP8) {
            char* dst = tiles + TILE_B + buf * RAW_B;
#pragma unroll
            for (int i = 0; i < D / 128; ++i) {            // raw image swizzled as in bank_fp8_fwd_kernel (fp8_swz)
                const int ii = w * (D / 128) + i;
                const int p = ii * 1024 + lane * 16;
                const int r = p / D, cp = (p % D) >> 4;
                const int c = cp ^ fp8_swz<D>(r);
                glds16(rs, dst + ii * 1024, (uint32_t)(mrow0 + r) * (uint32_t)D + (uint32_t)c * 16u);
            }
        } else {
            char* dst = tiles + buf * TILE_B;
#pragma unroll
            for (int i = 0; i < GLDS_PER_WAVE; ++i) {
                const int ii = w * GLDS_PER_WAVE + i;
                const int p = ii * 1024 + lane * 16;
                const int r = p / ROWB, cp = (p % ROWB) >> 4;
                const int c = cp ^ bank_swz(r & 15);
                glds16(rs, dst + ii * 1024, (uint32_t)(mrow0 + r) * (uint32_t)ROWB + (uint32_t)c * 16u);
            }
        }
    };
    // row scales of the thread's D / 128 raw pieces, loaded ONE TILE AHEAD (behind the tile's DMA): a global load where
    // the scale is needed puts an L2 / HBM round trip on every tile's critical path
    [[maybe_unused]] float sc_next[FP8 ? D / 128 : 1];
    [[maybe_unused]] float sc_row = 0.f;                 // F8L: scale of tile row tid (tid < TR), for the statistics
    auto load_scales = [&](int t) {
        const int mrow0 = m_lo + t * TR;
#pragma unroll
        for (int i = 0; i < D / 128; ++i) {
            const int r = ((i * 256 + tid) * 16) / D;
            sc_next[i] = (mrow0 + r < a.M) ? a.bank_scale[mrow0 + r] : 0.f;
        }
        if constexpr (F8L) {
            if (tid < TR) sc_row = (mrow0 + tid < a.M) ? a.bank_scale[mrow0 + tid] : 0.f;
        }
    };
    auto dequant = [&](int t, int buf, const float (&scv)[FP8 ? D / 128 : 1]) {
        const char* raw = tiles + TILE_B + buf * RAW_B;
#pragma unroll
        for (int i = 0; i < D / 128; ++i) {
            const int p = (i * 256 + tid) * 16;          // byte offset of this thread's 16 e4m3 values in the raw image
            const int r = p / D, cb = ((p % D) >> 4) ^ fp8_swz<D>(r);        // logical 16-value chunk at that position
            const float sc = scv[i];
            const u32x4 v = *(const u32x4*)(raw + p);
            bf16x8 o[2];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[d], false);
                const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[d], true);
                o[d >> 1][(d & 1) * 4 + 0] = f2bf(lo[0] * sc);
                o[d >> 1][(d & 1) * 4 + 1] = f2bf(lo[1] * sc);
                o[d >> 1][(d & 1) * 4 + 2] = f2bf(hi[0] * sc);
                o[d >> 1][(d & 1) * 4 + 3] = f2bf(hi[1] * sc);
            }
            const int sw = bank_swz(r & 15);
            *(bf16x8*)(tiles + r * ROWB + (((2 * cb) ^ sw) << 4)) = o[0];
            *(bf16x8*)(tiles + r * ROWB + (((2 * cb + 1) ^ sw) << 4)) = o[1];
        }
    };

    constexpr int NDMA = D / 128;                        // e4m3: DMA instructions per wave and raw tile
    if (ntiles > 0) {
        if constexpr (FP8) {                             // scales before the DMA they travel with: the counted wait below
            load_scales(0);                              // leaves exactly the LAST tile's DMA in flight
            asm volatile("" ::: "memory");
        }
        stage(0, 0);
        if constexpr (RS == 3) {
            if (ntiles > 1) stage(1, 1);
        }
    }
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        [[maybe_unused]] const int rbuf = RS == 3 ? t % 3 : buf;       // raw tile of an e4m3 bank
        if constexpr (RS == 3) {
            if (t + 1 < ntiles) wait_vmcnt<NDMA>();      // tile t and its scales are in, tile t + 1 may still be in flight
            else wait_vmcnt<0>();
        } else {
            wait_vm0();
        }
        lds_barrier();
        [[maybe_unused]] float sc_cur[FP8 ? D / 128 : 1];
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < D / 128; ++i) sc_cur[i] = sc_next[i];
            if constexpr (F8L) {
                if (tid < TR) Sst[buf * TR + tid] = sc_row;          // read after the Sp barrier below
            }
            asm volatile("" : "+v"(sc_cur[0]));          // the copies are taken before the next tile's loads are issued
            if (t + 1 < ntiles) load_scales(t + 1);
            asm volatile("" ::: "memory");
            if constexpr (RS == 3) {
                if (t + 2 < ntiles) stage(t + 2, (t + 2) % 3);       // raw[(t + 2) % 3] = raw[(t - 1) % 3]: consumed before this barrier
            } else {
                if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
            }
            if constexpr (!F8L) {
                dequant(t, rbuf, sc_cur);
                lds_barrier();
            }
        } else {
            if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
        }
        const char* T = tiles + (FP8 ? 0 : buf * TILE_B);

        // ---- logits over this wave's d-slice: D[i = key][j = query]
        f32x4 s[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) s[mt][nt] = f32x4{0, 0, 0, 0};
        if constexpr (F8L) {
            const char* R = tiles + TILE_B + rbuf * RAW_B;
            f32x4 sl4[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) sl4[mt][nt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int r = nt * 16 + (lane & 15);
                    const int kb = w * DW + ks * 32 + (lane >> 4) * 8;               // byte offset of this lane's 8 values
                    const long bfr = *(const long*)(R + r * D + (((kb >> 4) ^ fp8_swz<D>(r)) << 4) + (kb & 8));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        s[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr, qh[mt][ks], s[mt][nt], 0, 0, 0);
                        sl4[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr, ql[mt][ks], sl4[mt][nt], 0, 0, 0);
                    }
                }
            }
            dequant(t, rbuf, sc_cur);                    // VALU + LDS beside the MFMAs; the image is read after two barriers
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) s[mt][nt] += sl4[mt][nt] * 0.0625f;
        } else {
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int r = nt * 16 + (lane & 15);
                const int c = (w * DW + ks * 32) / 8 + (lane >> 4);
                const bf16x8 bfrag = *(const bf16x8*)(T + r * ROWB + ((c ^ bank_swz(r & 15)) << 4));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) s[mt][nt] = mfma16b(bfrag, qf[mt][ks], s[mt][nt]);
            }
        }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *(f32x4*)(Sp + (((w * 2 + mt) * 2 + nt) * 64 + lane) * 4) = s[mt][nt];
        lds_barrier();
        f32x4 sv = {0, 0, 0, 0};
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) sv += *(const f32x4*)(Sp + (((ww * 2 + mt_o) * 2 + nt_o) * 64 + lane) * 4);

        const int key0 = m_lo + t * TR + nt_o * 16 + (lane >> 4) * 4;   // shard-local row of sv[0]
        if constexpr (GRP) {
            // both key-half waves of a query row reduce the whole tile (the other half's partial sums are read
            // as well), so the max and its position are known to each without another barrier
            f32x4 so = {0, 0, 0, 0};
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) so += *(const f32x4*)(Sp + (((ww * 2 + mt_o) * 2 + (nt_o ^ 1)) * 64 + lane) * 4);
            float gm = -INFINITY;
            int gi = TR;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k_own = nt_o * 16 + (lane >> 4) * 4 + r, k_oth = (nt_o ^ 1) * 16 + (lane >> 4) * 4 + r;
                if (sv[r] > gm || (sv[r] == gm && k_own < gi)) { gm = sv[r]; gi = k_own; }
                if (so[r] > gm || (so[r] == gm && k_oth < gi)) { gm = so[r]; gi = k_oth; }
            }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const float m2 = __shfl_xor(gm, off, 64);
                const int i2 = __shfl_xor(gi, off, 64);
                if (m2 > gm || (m2 == gm && i2 < gi)) { gm = m2; gi = i2; }
            }
            const int tgt = m_lo / TR + t;                               // shard-local target id of this tile
            const float z = gm * a.inv_tau;
            if constexpr (!BWD) {
                if (nt_o == 0 && (lane >> 4) == 0) {                     // one lane per query row keeps the statistics
                    const float mn = fmaxf(st_m, z);
                    st_l = st_l * __expf(st_m - mn) + __expf(z - mn);
                    st_m = mn;
                    st_sl += z;
                    if ((int64_t)tgt == label) st_lab = z;
                }
            } else {
                float gv = 0.f;
                if (q_ok) {
                    gv = __expf(z - lse) - label_smoothing * inv_m_total;
                    if ((int64_t)tgt == label) gv -= 1.0f - label_smoothing;
                }
                bf16x4 gb;
#pragma unroll
                for (int r = 0; r < 4; ++r) gb[r] = f2bf(nt_o * 16 + (lane >> 4) * 4 + r == gi ? gv : 0.f);
                *(bf16x4*)(Gs + (mt_o * 16 + (lane & 15)) * LDG + nt_o * 16 + (lane >> 4) * 4) = gb;
            }
        } else if constexpr (!BWD) {
            float tm = -INFINITY;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = (key0 + r < a.M) ? sv[r] * a.inv_tau : -INFINITY;
                tm = fmaxf(tm, v[r]);
            }
            if (tm > -INFINITY) {
                const float mn = fmaxf(st_m, tm);
                float add = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (key0 + r < a.M) {
                        add += __expf(v[r] - mn);
                        st_sl += v[r];
                        if ((int64_t)(key0 + r) == label) st_lab = v[r];
                    }
                }
                st_l = st_l * __expf(st_m - mn) + add;
                st_m = mn;
            }
        } else if constexpr (FUSED) {
            // tile maximum of this lane's query over all 32 keys: both key-half waves read both halves' partial sums
            f32x4 so = {0, 0, 0, 0};
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) so += *(const f32x4*)(Sp + (((ww * 2 + mt_o) * 2 + (nt_o ^ 1)) * 64 + lane) * 4);
            const int key0o = m_lo + t * TR + (nt_o ^ 1) * 16 + (lane >> 4) * 4;
            float z[4], tmx = -INFINITY;
            [[maybe_unused]] f32x4 sb4 = {1.f, 1.f, 1.f, 1.f}, sbo = {1.f, 1.f, 1.f, 1.f};
            if constexpr (F8L) {                         // logit = row scale x query scale x (hi + lo / 16) / tau
                sb4 = *(const f32x4*)(Sst + buf * TR + nt_o * 16 + (lane >> 4) * 4) * zs;
                sbo = *(const f32x4*)(Sst + buf * TR + (nt_o ^ 1) * 16 + (lane >> 4) * 4) * zs;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z[r] = (key0 + r < a.M) ? (F8L ? sv[r] * sb4[r] : sv[r] * a.inv_tau) : -INFINITY;
                const float zo = (key0o + r < a.M) ? (F8L ? so[r] * sbo[r] : so[r] * a.inv_tau) : -INFINITY;
                tmx = fmaxf(tmx, fmaxf(z[r], zo));
            }
            tmx = fmaxf(tmx, __shfl_xor(tmx, 16, 64));
            tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
            // The reference exponent st_m is an INTEGER power of two (log2 domain) and p = 2^(z log2e - st_m) is built as
            // exp2(fraction) scaled by an exact ldexp: bf16(p) = 2^-st_m bf16(2^(z log2e)) whatever st_m is, and a change of
            // reference rescales the accumulators by an exact power of two.  So the rounding of every p - and with it dq up
            // to fp32 summation order - does not depend on how the bank is cut into chunks, shards or query blocks: a
            // data-parallel step reproduces the single-process one as closely as the two-pass kernels do.
            // Lazy: st_m moves only when a tile exceeds it by more than 2^FUSED_SLACK (p <= 2^12, harmless in bf16 / fp32);
            // with 32 queries per block SOME row sets a new record in almost every tile of a short chunk.
            constexpr float FUSED_SLACK = 12.0f, LOG2E = 1.4426950408889634f;
            const float tm2 = tmx * LOG2E;
            const float mn = (tm2 > st_m + FUSED_SLACK) ? ceilf(tm2) : st_m;    // st_m = -inf: the first live tile sets it
            const float alpha = (mn > st_m && st_m > -INFINITY) ? ldexpf(1.0f, (int)(st_m - mn)) : 1.0f;   // nothing accumulated before the first live tile
            float add = 0.f;
            bf16x4 gb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = z[r] > -INFINITY;
                const float z2 = z[r] * LOG2E, zf = floorf(z2);
                const float pv = live ? ldexpf(__builtin_amdgcn_exp2f(z2 - zf), (int)fmaxf(zf - mn, -200.0f)) : 0.f;
                add += pv;
                st_sl += live ? z[r] : 0.f;
                const bool is_lab = live && (int64_t)(key0 + r) == label;
                st_lab = is_lab ? z[r] : st_lab;
                // the label key stays out of O: the fold subtracts (1 - p_label) bank_label with p_label in fp32, so a
                // confident row (p_label -> 1) does not turn into the difference of two bf16-rounded near-equal sums
                gb[r] = f2bf((q_ok && !is_lab) ? pv : 0.f);
            }
            st_l = st_l * alpha + add;
            st_m = mn;
            *(bf16x4*)(Gs + (mt_o * 16 + (lane & 15)) * LDG + nt_o * 16 + (lane >> 4) * 4) = gb;
            if (nt_o == 0 && lane < 16) As[mt_o * 16 + lane] = alpha;
        } else {
            bf16x4 gb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gv = 0.f;
                if (q_ok && key0 + r < a.M) {
                    gv = __expf(sv[r] * a.inv_tau - lse) - label_smoothing * inv_m_total;
                    if ((int64_t)(key0 + r) == label) gv -= 1.0f - label_smoothing;
                }
                gb[r] = f2bf(gv);
            }
            *(bf16x4*)(Gs + (mt_o * 16 + (lane & 15)) * LDG + nt_o * 16 + (lane >> 4) * 4) = gb;
        }
        if constexpr (BWD) {
            lds_barrier();
            if constexpr (FUSED) {                       // the accumulators follow the row maximum (lane: query lane & 15 of each mt)
                const float a0 = As[lane & 15], a1 = As[16 + (lane & 15)];
                if (__any(a0 != 1.0f || a1 != 1.0f)) {   // the maximum moves in the first tiles of a chunk, then rarely
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) { dq[0][dt] *= a0; dq[1][dt] *= a1; }
                }
            }
            // ---- dq[q][d] += sum_key G[q][key] bank[key][d]:  D[i = d][j = query], k = key (one 32-step)
            bf16x8 gf[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                gf[mt] = *(const bf16x8*)(Gs + (mt * 16 + (lane & 15)) * LDG + (lane >> 4) * 8);
            // transposed fragments through the asm read (no vmcnt(0) in front of them, common.h), in two halves: the MFMAs of
            // the first run while the second half's reads return
            union TF { s16x4 h[2]; bf16x8 v; } u[NDT];
            constexpr int H1 = (NDT + 1) / 2;
            lds_tie(gf[0]);                                  // the G fragments are read (and waited for) before the asm reads
            lds_tie(gf[1]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int cb = w * DW + dt * 16;   // first column of this d tile
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
                    const int col = cb + (lane & 3) * 4;
                    u[dt].h[h] = lds_tr16_b64_asm(T + r * ROWB + (((col >> 3) ^ bank_swz(r & 15)) << 4) + (col & 7) * 2);
                }
            }
            wait_lgkm<2 * (NDT - H1)>();
#pragma unroll
            for (int dt = 0; dt < H1; ++dt) { lds_tie(u[dt].h[0]); lds_tie(u[dt].h[1]); }
#pragma unroll
            for (int dt = 0; dt < H1; ++dt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) dq[mt][dt] = mfma16b(u[dt].v, gf[mt], dq[mt][dt]);
            wait_lgkm<0>();
#pragma unroll
            for (int dt = H1; dt < NDT; ++dt) { lds_tie(u[dt].h[0]); lds_tie(u[dt].h[1]); }
#pragma unroll
            for (int dt = H1; dt < NDT; ++dt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) dq[mt][dt] = mfma16b(u[dt].v, gf[mt], dq[mt][dt]);
        }
    }

    if constexpr (!BWD) {
        // combine the 4 lane groups (lane>>4) and the 2 key-half waves of each query row
        __syncthreads();
        float* f = Fin + ((w * 64 + lane) * 4);
        f[0] = st_m; f[1] = st_l; f[2] = st_sl; f[3] = st_lab;
        __syncthreads();
        if (tid < BQ) {
            const int mt = tid >> 4, ql = tid & 15;
            float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
            for (int nt = 0; nt < 2; ++nt)
                for (int g = 0; g < 4; ++g) {
                    const float* p = Fin + (((mt * 2 + nt) * 64 + g * 16 + ql) * 4);
                    const float mn = fmaxf(m, p[0]);
                    if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
                    m = mn;
                    sl += p[2];
                    lab = fmaxf(lab, p[3]);
                }
            const int q = q0 + tid;
            if (q < a.B) {
                float* o = ws + ((size_t)mi * a.B + q) * 4;
                o[0] = m; o[1] = l; o[2] = sl; o[3] = lab;
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int q = q0 + mt * 16 + (lane & 15);
            if (q >= a.B) continue;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                *(f32x4*)(ws + ((size_t)mi * a.B + q) * D + w * DW + dt * 16 + (lane >> 4) * 4) = dq[mt][dt];
            }
        }
        if constexpr (FUSED) {
            // statistics of the chunk: every lane group / key-half wave of a query shares the running maximum, so l, sum z
            // add up and the label logit is the maximum of the (at most one) finite entry
            __syncthreads();
            float* f = Fin + ((w * 64 + lane) * 4);
            f[0] = st_m; f[1] = st_l; f[2] = st_sl; f[3] = st_lab;
            __syncthreads();
            if (tid < BQ) {
                const int mt = tid >> 4, ql = tid & 15;
                float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
                for (int nt = 0; nt < 2; ++nt)
                    for (int g = 0; g < 4; ++g) {
                        const float* p = Fin + (((mt * 2 + nt) * 64 + g * 16 + ql) * 4);
                        m = fmaxf(m, p[0]);
                        l += p[1];
                        sl += p[2];
                        lab = fmaxf(lab, p[3]);
                    }
                const int q = q0 + tid;
                if (q < a.B) {
                    float* o = ws2 + ((size_t)mi * a.B + q) * 4;
                    o[0] = m * 0.6931471805599453f; o[1] = l; o[2] = sl; o[3] = lab;    // the reference exponent in natural-log units
                }
            }
            // arrival counter of bank_step_tail_kernel (the word behind the statistics partials): reset by the launch in front of it
            if (blockIdx.x == 0 && tid == 0) *(int*)(ws2 + (size_t)ck.nchunks * a.B * 4) = 0;
        }
    }
}

// dq[b, :] = alpha * ( sum_c exp(m_c[b] - lse[b]) O_c[b, :]  -  (1 - p_label) bank[label_b, :] if the label lies in this shard )
// (O excludes the label key; p_label = exp(z_label - lse) from the chunk statistics, the same fp32 arithmetic as
// softmax - onehot): the "backward" half of the fused pass - no bank traffic but the B label rows.
// The partials stay fp32: rounding them would make dq depend on where the chunk boundaries fall (see the kernel).
// Block = (64 columns, one query): 16 column quads x 16 chunk groups.
template <bool FP8>
__global__ __launch_bounds__(256) void bank_fused_combine_kernel(const float* __restrict__ Op, const float* __restrict__ sp, int nch,
                                                                int B, int D, const float* __restrict__ lse,
                                                                const void* __restrict__ bank, const float* __restrict__ scale,
                                                                const int64_t* __restrict__ labels, int m_begin, int M,
                                                                float alpha, float* __restrict__ dq, int lddq) {
    __shared__ float red[16][16][5];
    __shared__ float zl[16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int b = blockIdx.y, c = (blockIdx.x * 16 + cq) * 4;
    const float ls = lse[b];
    f32x4 s = {0, 0, 0, 0};
    float zlab = -INFINITY;
    const bool in = c < D;
    const float* o = Op + (size_t)b * D + (in ? c : 0);
    const float* st = sp + (size_t)b * 4;
    const size_t so = (size_t)B * D, ss = (size_t)B * 4;
    int r = rl;
    for (; r + 112 < nch; r += 128) {                                // eight independent (weight, partial) loads in flight; the
        f32x4 v[8];                                                  // accumulation order is that of the 4-deep loop below
        float mm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = *(const f32x4*)(o + (size_t)(r + 16 * k) * so);
            mm[k] = st[(size_t)(r + 16 * k) * ss];
            zlab = fmaxf(zlab, st[(size_t)(r + 16 * k) * ss + 3]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k] * __expf(mm[k] - ls);
    }
    for (; r + 48 < nch; r += 64) {                                  // four independent (weight, partial) loads in flight
        const f32x4 v0 = *(const f32x4*)(o + (size_t)r * so), v1 = *(const f32x4*)(o + (size_t)(r + 16) * so);
        const f32x4 v2 = *(const f32x4*)(o + (size_t)(r + 32) * so), v3 = *(const f32x4*)(o + (size_t)(r + 48) * so);
        const float* s0 = st + (size_t)r * ss;
        const float m0 = s0[0], m1 = s0[16 * ss], m2 = s0[32 * ss], m3 = s0[48 * ss];
        zlab = fmaxf(fmaxf(zlab, s0[3]), fmaxf(fmaxf(s0[16 * ss + 3], s0[32 * ss + 3]), s0[48 * ss + 3]));
        s += v0 * __expf(m0 - ls);                                   // chunk reference against the global lse
        s += v1 * __expf(m1 - ls);
        s += v2 * __expf(m2 - ls);
        s += v3 * __expf(m3 - ls);
    }
    for (; r < nch; r += 16) {
        const float* s0 = st + (size_t)r * ss;
        zlab = fmaxf(zlab, s0[3]);
        s += *(const f32x4*)(o + (size_t)r * so) * __expf(s0[0] - ls);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cq][e] = s[e];
    if (cq == 0) zl[rl] = zlab;
    __syncthreads();
    if (threadIdx.x < 64) {                                          // thread = one column of the block
        const int col = blockIdx.x * 64 + threadIdx.x;
        float t = 0.f, z = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t += red[k][threadIdx.x >> 2][threadIdx.x & 3]; z = fmaxf(z, zl[k]); }
        if (col < D) {
            const int64_t lab = labels[b] - (int64_t)m_begin;
            if (lab >= 0 && lab < (int64_t)M) {
                const float cl = 1.0f - __expf(z - ls);
                float v;
                if constexpr (FP8) {
                    const uint32_t w4 = *(const uint32_t*)((const uint8_t*)bank + (size_t)lab * D + (col & ~3));
                    const f32x2 p2 = (col & 2) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)w4, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)w4, false);
                    v = bf2f(f2bf(p2[col & 1] * scale[lab]));        // the kernels see the dequantised row rounded to bf16
                } else {
                    v = bf2f(((const bf16_t*)bank)[(size_t)lab * D + col]);
                }
                t -= v * cl;
            }
            dq[(size_t)b * lddq + col] = t * alpha;
        }
    }
}

// The whole tail of a single-shard step in ONE launch (bank_step): fold of the chunk statistics (what bank_stats_fold_kernel +
// bank_loss_finalize_kernel do in two launches of a handful of workgroups - 4 us each, launch-bound), the fold of the chunk
// partials with the label row subtracted (bank_fused_combine_kernel), row_lse / row_loss, and the mean loss.  Block =
// (64 columns, one query) as in the combine kernel; every block folds its query's nch x {m, l, sum z, label z} itself
// (4 KB from L2 - cheaper than a launch boundary).  The mean is summed in a FIXED order by the wave that arrives last
// (ticket on a counter the pass in front resets; release / acquire at agent scope): bit-reproducible, no atomics on floats.
template <bool FP8>
__global__ __launch_bounds__(256) void bank_step_tail_kernel(const float* __restrict__ Op, const float* __restrict__ sp, int nch,
                                                            int B, int D, const void* __restrict__ bank, const float* __restrict__ scale,
                                                            const int64_t* __restrict__ labels, int M, float alpha,
                                                            float* __restrict__ dq, int lddq, float* __restrict__ row_lse,
                                                            float* row_loss, float* __restrict__ loss_mean, int* counter) {
    __shared__ float red[16][16][5];
    __shared__ float fm[1], fl[1], fz[1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cq = tid & 15, rl = tid >> 4;
    const int b = blockIdx.y, c = (blockIdx.x * 16 + cq) * 4;
    const size_t so = (size_t)B * D, ss = (size_t)B * 4;
    const float* st = sp + (size_t)b * 4;
    // ---- statistics of query b over all chunks: wave 0 repeats bank_stats_fold_kernel's arithmetic (lanes stride 64, the same
    // merge order) and bank_loss_finalize_kernel's single-shard lse, so that lse - hence every p and dq - is BIT-identical to
    // the three-call path (a sharded data-parallel step then reproduces the single-process one as before)
    if (w == 0) {
        float m = -INFINITY, l = 0.f, zlab = -INFINITY;
        for (int i = lane; i < nch; i += 64) {
            const f32x4 p = *(const f32x4*)(st + (size_t)i * ss);
            const float mn = fmaxf(m, p[0]);
            if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
            m = mn;
            zlab = fmaxf(zlab, p[3]);
        }
        const float mw = wave_max(m);
        l = (m > -INFINITY) ? l * __expf(m - mw) : 0.f;
        l = wave_sum(l);
        zlab = wave_max(zlab);
        if (lane == 0) { fm[0] = mw; fl[0] = l; fz[0] = zlab; }
    }
    __syncthreads();
    const float ls = fm[0] + logf(fl[0]);
    const float z = fz[0];
    // ---- dq[b, 64 columns] = alpha (sum_c exp(m_c - lse) O_c - (1 - p_label) bank[label])
    f32x4 s = {0, 0, 0, 0};
    const bool in = c < D;
    const float* o = Op + (size_t)b * D + (in ? c : 0);
    int r = rl;
    for (; r + 112 < nch; r += 128) {                                // as bank_fused_combine_kernel: same order, eight loads in flight
        f32x4 v[8];
        float mm[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = *(const f32x4*)(o + (size_t)(r + 16 * k) * so);
            mm[k] = st[(size_t)(r + 16 * k) * ss];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k] * __expf(mm[k] - ls);
    }
    for (; r + 48 < nch; r += 64) {
        const f32x4 v0 = *(const f32x4*)(o + (size_t)r * so), v1 = *(const f32x4*)(o + (size_t)(r + 16) * so);
        const f32x4 v2 = *(const f32x4*)(o + (size_t)(r + 32) * so), v3 = *(const f32x4*)(o + (size_t)(r + 48) * so);
        const float* s0 = st + (size_t)r * ss;
        const float m0 = s0[0], m1 = s0[16 * ss], m2 = s0[32 * ss], m3 = s0[48 * ss];
        s += v0 * __expf(m0 - ls);
        s += v1 * __expf(m1 - ls);
        s += v2 * __expf(m2 - ls);
        s += v3 * __expf(m3 - ls);
    }
    for (; r < nch; r += 16) s += *(const f32x4*)(o + (size_t)r * so) * __expf(st[(size_t)r * ss] - ls);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][cq][e] = s[e];
    __syncthreads();
    if (tid < 64) {                                                  // thread = one column of the block
        const int col = blockIdx.x * 64 + tid;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tid >> 2][tid & 3];
        if (col < D) {
            const int64_t lab = labels[b];
            if (lab >= 0 && lab < (int64_t)M) {
                const float cl = 1.0f - __expf(z - ls);
                float v;
                if constexpr (FP8) {
                    const uint32_t w4 = *(const uint32_t*)((const uint8_t*)bank + (size_t)lab * D + (col & ~3));
                    const f32x2 p2 = (col & 2) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)w4, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)w4, false);
                    v = bf2f(f2bf(p2[col & 1] * scale[lab]));
                } else {
                    v = bf2f(((const bf16_t*)bank)[(size_t)lab * D + col]);
                }
                t -= v * cl;
            }
            dq[(size_t)b * lddq + col] = t * alpha;
        }
        // ---- per-row outputs and the mean loss (the blocks of column group 0, wave 0)
        if (blockIdx.x == 0) {
            int ticket = 0;
            if (tid == 0) {
                row_lse[b] = ls;
                row_loss[b] = ls - z;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            ticket = __shfl(ticket, 0, 64);
            if (ticket == B - 1) {                                   // every row_loss is published: sum them in index order
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                float acc = 0.f;
                for (int i = tid; i < B; i += 64) acc += __hip_atomic_load(row_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc = wave_sum(acc);
                if (tid == 0) *loss_mean = acc / (float)B;
            }
        }
    }
}

// ------------------------------------------------------------------ fp8 bank, forward pass on the fp8 MFMA
// BASELINE config 5 ("fp8 MFMA sim-matmul"): the e4m3 bank tile goes from HBM to LDS (1 byte per element) and from LDS
// straight into v_mfma_f32_16x16x32_fp8_fp8 - no dequantised bf16 image, no extra LDS pass or barrier.  The MFMA needs
// both operands in fp8, so the block's 32 bf16 queries are split once, in registers, into TWO e4m3 terms with one scale
// per query:  q ~= sq * hi + (sq / 16) * lo,  sq = max|q| / 448, hi = e4m3(q * (1 / sq)), lo = e4m3((q - sq hi) * (16 / sq))
// (|q - sq hi| <= half an ulp of hi <= 16 sq, so lo never saturates; the pair carries q to ~2^-8, the bf16 level).
// logit = sb[key] * sq * (acc_hi + acc_lo / 16) / tau.  Same block / chunk geometry and statistics as bank_stream_kernel.
// LDS image of the raw tile [TR rows][D bytes]: 16-byte chunk c of row r sits at position c ^ x(r) of its aligned
// group - x = r & 15 when a row is a multiple of 256 B (every row starts on bank 0), (r >> 1) & 7 when it is an odd
// multiple of 128 B (rows alternate between two bank halves): the 16 rows of a fragment read fall in 16 distinct
// 16-byte bank groups.
static constexpr int FP8_MAX_CHUNK_ROWS = 2048;   // rows of a block's chunk whose scales are kept in LDS (8 KB)
static constexpr int FP8_STAGES = 4;     // raw tiles in flight per block: a chunk is ~5 tiles, so nearly all of it is
                                         // requested at once (two stages: one DMA latency per tile, 22.9 us at B = 32)
template <int D>
__global__ __launch_bounds__(256, 1) void bank_fp8_fwd_kernel(BankArgs a, BankChunking ck, float* __restrict__ ws) {
    constexpr int DW = D / 4;            // columns per wave
    constexpr int KSW = DW / 32;         // 32-deep k-steps per wave
    constexpr int RAW_B = TR * D;        // bytes of one raw tile
    constexpr int S = FP8_STAGES, NDMA = D / 128;         // DMA instructions per wave and tile
    static_assert(D % 128 == 0 && RAW_B % 1024 == 0 && (S - 1) * NDMA <= 63, "bank width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tiles = smem;                                   // S x RAW_B
    float* Sp = (float*)(smem + S * RAW_B);               // [4 waves][2 mt][2 nt][64 lanes][4]; reused as Fin at the end
    float* Qm = (float*)(smem + S * RAW_B + 4 * 4096);    // [4 waves][32 queries] partial max |q|
    float* Ssc = Qm + 4 * 32;                             // per-row scales of the block's chunk (<= FP8_MAX_CHUNK_ROWS)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mi = blockIdx.x % ck.nchunks, qi = blockIdx.x / ck.nchunks;
    const int q0 = qi * BQ;
    const int m_lo = mi * ck.rows;
    const int m_hi = min(a.M, m_lo + ck.rows);
    const int ntiles = m_hi > m_lo ? (m_hi - m_lo + TR - 1) / TR : 0;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bank, (uint32_t)a.M * (uint32_t)D);

    auto stage = [&](int t, int buf) {
        const int mrow0 = m_lo + t * TR;
        char* dst = tiles + buf * RAW_B;
#pragma unroll
        for (int i = 0; i < D / 128; ++i) {
            const int ii = w * (D / 128) + i;
            const int p = ii * 1024 + lane * 16;
            const int r = p / D, cp = (p % D) >> 4;
            const int c = cp ^ fp8_swz<D>(r);
            glds16(rs, dst + ii * 1024, (uint32_t)(mrow0 + r) * (uint32_t)D + (uint32_t)c * 16u);
        }
    };
    // the chunk's row scales into LDS once (a global load where a scale is needed costs the tile an L2 round trip, and
    // VMEM loads inside the loop would disturb the counted waits of the tile ring); visible after the barrier below
    for (int i = tid; i < ntiles * TR; i += 256) Ssc[i] = a.bank_scale[min(m_lo + i, a.M - 1)];
    // ---- this wave's d-slice of the 32 queries -> two e4m3 terms (B operand: j = query, k = d)
    long qh[2][KSW], ql[2][KSW];
    float sq[2];
    {
        bf16x8 qf[2][KSW];
        float am[2] = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int qr = q0 + mt * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                if (qr < a.B) {
                    qf[mt][ks] = *(const bf16x8*)(a.q + (size_t)qr * a.ldq + w * DW + ks * 32 + (lane >> 4) * 8);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[mt][ks][e] = (bf16_t)0.0f;
                }
            }
        }
        // the first S-1 tiles go out behind the query loads (vmcnt retires in order: the split below waits for the
        // queries only) and land while the queries are split
#pragma unroll
        for (int t = 0; t < S - 1; ++t)
            if (t < ntiles) stage(t, t);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) am[mt] = fmaxf(am[mt], fabsf(bf2f(qf[mt][ks][e])));
            am[mt] = fmaxf(am[mt], __shfl_xor(am[mt], 16, 64));
            am[mt] = fmaxf(am[mt], __shfl_xor(am[mt], 32, 64));
            if ((lane >> 4) == 0) Qm[w * 32 + mt * 16 + lane] = am[mt];
        }
        lds_barrier();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float m = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) m = fmaxf(m, Qm[ww * 32 + mt * 16 + (lane & 15)]);
            sq[mt] = m > 0.f ? m / 448.0f : 1.0f;
            // one division per query; every element is scaled by the reciprocal (192 fp32 divisions per lane cost 4 us).
            // Separate multiply and subtract (no fma contraction): oracle/bank_loss.py split_query_e4m3 does the same.
            const float rh = 1.0f / sq[mt], rl = rh * 16.0f;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                float v[8], hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(__fmul_rn(bf2f(qf[mt][ks][e]), rh), -448.0f), 448.0f);
                qh[mt][ks] = pack_fp8x8(v);
                unpack_fp8x8(qh[mt][ks], hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float res = __fsub_rn(bf2f(qf[mt][ks][e]), __fmul_rn(hv[e], sq[mt]));
                    v[e] = fminf(fmaxf(__fmul_rn(res, rl), -448.0f), 448.0f);
                }
                ql[mt][ks] = pack_fp8x8(v);
            }
        }
    }

    // softmax-phase ownership as in bank_stream_kernel
    const int mt_o = w >> 1, nt_o = w & 1;
    const int q_o = q0 + mt_o * 16 + (lane & 15);
    const bool q_ok = q_o < a.B;
    const int64_t label = q_ok ? a.labels[q_o] - (int64_t)a.m_begin : -1;
    const float zs = sq[mt_o] * a.inv_tau;                // query scale x 1 / tau
    float st_m = -INFINITY, st_l = 0.f, st_sl = 0.f, st_lab = -INFINITY;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t % S;
        if (t + S - 1 < ntiles) stage(t + S - 1, (t + S - 1) % S);    // into the buffer of tile t-1 (all waves are past it)
        const int behind = min(S - 1, ntiles - 1 - t);               // tiles requested after tile t
        if (behind >= 3) wait_vmcnt<3 * NDMA>();
        else if (behind == 2) wait_vmcnt<2 * NDMA>();
        else if (behind == 1) wait_vmcnt<NDMA>();
        else wait_vmcnt<0>();
        lds_barrier();
        const char* T = tiles + buf * RAW_B;
        f32x4 sh[2][2], sl4[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sh[mt][nt] = f32x4{0, 0, 0, 0};
                sl4[mt][nt] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int r = nt * 16 + (lane & 15);
                const int kb = w * DW + ks * 32 + (lane >> 4) * 8;               // byte offset of this lane's 8 values
                const long bfr = *(const long*)(T + r * D + (((kb >> 4) ^ fp8_swz<D>(r)) << 4) + (kb & 8));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    sh[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr, qh[mt][ks], sh[mt][nt], 0, 0, 0);
                    sl4[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr, ql[mt][ks], sl4[mt][nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                *(f32x4*)(Sp + (((w * 2 + mt) * 2 + nt) * 64 + lane) * 4) = sh[mt][nt] + sl4[mt][nt] * 0.0625f;
        lds_barrier();
        f32x4 sv = {0, 0, 0, 0};
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) sv += *(const f32x4*)(Sp + (((ww * 2 + mt_o) * 2 + nt_o) * 64 + lane) * 4);
        const int key0 = m_lo + t * TR + nt_o * 16 + (lane >> 4) * 4;           // shard-local row of sv[0]
        const f32x4 sb4 = *(const f32x4*)(Ssc + t * TR + nt_o * 16 + (lane >> 4) * 4);
        float tm = -INFINITY;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = (key0 + r < a.M) ? sv[r] * sb4[r] * zs : -INFINITY;
            tm = fmaxf(tm, v[r]);
        }
        if (tm > -INFINITY) {
            const float mn = fmaxf(st_m, tm);
            float add = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = key0 + r < a.M;
                add += live ? __expf(v[r] - mn) : 0.f;
                st_sl += live ? v[r] : 0.f;
                st_lab = (live && (int64_t)(key0 + r) == label) ? v[r] : st_lab;
            }
            st_l = st_l * __expf(st_m - mn) + add;
            st_m = mn;
        }
    }
    // combine the 4 lane groups (lane>>4) and the 2 key-half waves of each query row
    __syncthreads();
    float* Fin = Sp;
    float* f = Fin + ((w * 64 + lane) * 4);
    f[0] = st_m; f[1] = st_l; f[2] = st_sl; f[3] = st_lab;
    __syncthreads();
    if (tid < BQ) {
        const int mt = tid >> 4, ql_ = tid & 15;
        float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
        for (int nt = 0; nt < 2; ++nt)
            for (int g = 0; g < 4; ++g) {
                const float* p = Fin + (((mt * 2 + nt) * 64 + g * 16 + ql_) * 4);
                const float mn = fmaxf(m, p[0]);
                if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
                m = mn;
                sl += p[2];
                lab = fmaxf(lab, p[3]);
            }
        const int q = q0 + tid;
        if (q < a.B) {
            float* o = ws + ((size_t)mi * a.B + q) * 4;
            o[0] = m; o[1] = l; o[2] = sl; o[3] = lab;
        }
    }
}

template <int D>
static int launch_bank_fp8_fwd(const BankArgs& a, const BankChunking& c, float* ws, hipStream_t st) {
    const size_t lds = FP8_STAGES * (size_t)TR * D + 4 * 4096 + 4 * 32 * sizeof(float) + FP8_MAX_CHUNK_ROWS * sizeof(float);
    auto kern = bank_fp8_fwd_kernel<D>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        const double bytes = (double)a.M * D + 4.0 * a.M + (double)a.B * D * 2 + (double)a.B * 16;
        ProfScope prof(PK_BANK_FWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(c.nq * c.nchunks), dim3(256), lds, st, a, c, ws);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------ fp8 bank, fused pass entirely on the fp8 MFMA
// The fused single pass (bank_stream_kernel<.., FUSED>) over an e4m3 bank WITHOUT a dequantised bf16 image: logits as in
// bank_fp8_fwd_kernel (raw tile x two-term e4m3 queries), and the dq GEMM  dq[q][d] += sum_key G[q][key] raw[key][d]  on
// v_mfma_f32_16x16x32_fp8_bf8:
//   A = the raw tile, transposed on the way out of LDS by ds_read_b64_tr_b8 (lane s of a 16-lane group hands in the address
//       of row s >> 1, byte half s & 1 of a 16-column block; lane o receives column o of those 8 rows - probed on gfx950);
//   B = G = p x (row scale / c2) x 8 as THREE bf8 (e5m2) terms t0 + t1 + t2 (each the bf8 rounding of what the previous
//       ones left: 3 significant bits per term, ~2^-9 together - the accuracy of a bf16 G).  e5m2's 32 binades take the
//       whole range of p against the lazy power-of-two reference (p <= 2^12 -> G <= 2^15; entries 19 binades below the
//       reference flush to zero, < 2e-6 of the row's largest term each), so the three products accumulate straight into
//       the dq registers - no per-tile scale, no second accumulator set.  c2 = the chunk's largest row scale rounded up to
//       a power of two (exact to divide by, exact to multiply back in the epilogue).
// No bf16 image means no dequantisation pass (a quarter of the F8L kernel's time) and 77 KB of LDS per workgroup: two of them
// fit a CU (below).
static constexpr int FP8F_SLACK = 12;
// Two workgroups per CU (256 registers per wave, LDS within 80 KB: two or three raw stages): a tile costs a workgroup ~6 000
// cycles of barrier-separated phases (cycle stamps: 390 DMA issue, 1 480 logits, 1 370 statistics, 1 910 dq, ~900 in
// barriers) whatever the ring depth, so a second workgroup fills the other's stalls: 141 -> 103 us at 16 x 400 000 x 768 with
// 512 chunks; with 256 chunks (one workgroup per CU) the 250-register code is no slower than the 354-register, five-stage
// one (131.7 vs 141.3 us).  D = 1 024 keeps one workgroup per CU (two 32 KB stages + 28 KB).
template <int D>
static constexpr int fp8f_stages() {
    constexpr int rest = 4 * 4096 + 4 * BQ * 4 + FP8_MAX_CHUNK_ROWS * 4 + 3 * BQ * TR + BQ * 4 + 64;
    constexpr int s = (80 * 1024 - rest) / (TR * D);
    return s > 5 ? 5 : (s < 2 ? 2 : s);
}

// GT = number of e5m2 terms that carry G: 3 (~2^-9, a bf16 G) or 2 (6 significant bits: 2^-7, still 8x finer than the e4m3
// bank values G multiplies; 24 of a tile's 120 MFMAs per wave and a third of the term arithmetic less - SPN_BANK_FP8_GTERMS)
template <int D, int GT>
__global__ __launch_bounds__(256, 2) void bank_fp8_fused_kernel(BankArgs a, BankChunking ck, float* __restrict__ Op,
                                                               float* __restrict__ sp) {
    static_assert(GT == 2 || GT == 3, "G terms");
    constexpr int DW = D / 4, KSW = DW / 32, NDT = DW / 16, RAW_B = TR * D;
    constexpr int S = fp8f_stages<D>(), NDMA = D / 128;
    static_assert(S >= 2 && (S - 1) * NDMA <= 63 && D % 128 == 0, "bank width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tiles = smem;                                   // S x RAW_B
    float* Sp = (float*)(smem + S * RAW_B);               // [4 waves][2 mt][2 nt][64 lanes][4]; reused as Fin at the end
    float* Qm = Sp + 4096;                                // [4 waves][32 queries] partial max |q|; then the block maximum of the scales
    float* Ssc = Qm + 4 * BQ;                             // row scales of the chunk (<= FP8_MAX_CHUNK_ROWS)
    uint8_t* G8 = (uint8_t*)(Ssc + FP8_MAX_CHUNK_ROWS);   // [3 terms][BQ queries][TR keys] bf8
    float* As = (float*)(G8 + 3 * BQ * TR);               // rescale factor per query
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int mi = blockIdx.x % ck.nchunks, qi = blockIdx.x / ck.nchunks;
    const int q0 = qi * BQ;
    const int m_lo = mi * ck.rows;
    const int m_hi = min(a.M, m_lo + ck.rows);
    const int ntiles = m_hi > m_lo ? (m_hi - m_lo + TR - 1) / TR : 0;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bank, (uint32_t)a.M * (uint32_t)D);

    auto stage = [&](int t) {
        const int mrow0 = m_lo + t * TR;
        char* dst = tiles + (t % S) * RAW_B;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int ii = w * NDMA + i;
            const int p = ii * 1024 + lane * 16;
            const int r = p / D, cp = (p % D) >> 4;
            const int c = cp ^ fp8_swz<D>(r);
            glds16(rs, dst + ii * 1024, (uint32_t)(mrow0 + r) * (uint32_t)D + (uint32_t)c * 16u);
        }
    };
    float smax = 0.f;
    for (int i = tid; i < ntiles * TR; i += 256) {
        const float v = (m_lo + i < a.M) ? a.bank_scale[m_lo + i] : 0.f;
        Ssc[i] = v;
        smax = fmaxf(smax, v);
    }
    // ---- this wave's d-slice of the 32 queries -> two e4m3 terms (as bank_fp8_fwd_kernel)
    long qh[2][KSW], ql[2][KSW];
    float sq[2];
    {
        bf16x8 qf[2][KSW];
        float am[2] = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int qr = q0 + mt * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                if (qr < a.B) {
                    qf[mt][ks] = *(const bf16x8*)(a.q + (size_t)qr * a.ldq + w * DW + ks * 32 + (lane >> 4) * 8);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[mt][ks][e] = (bf16_t)0.0f;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < S - 1; ++t)
            if (t < ntiles) stage(t);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) smax = fmaxf(smax, __shfl_xor(smax, off, 64));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) am[mt] = fmaxf(am[mt], fabsf(bf2f(qf[mt][ks][e])));
            am[mt] = fmaxf(am[mt], __shfl_xor(am[mt], 16, 64));
            am[mt] = fmaxf(am[mt], __shfl_xor(am[mt], 32, 64));
            if ((lane >> 4) == 0) Qm[w * 32 + mt * 16 + lane] = am[mt];
        }
        if (lane == 0) As[w] = smax;
        __syncthreads();
        smax = fmaxf(fmaxf(As[0], As[1]), fmaxf(As[2], As[3]));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float m = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) m = fmaxf(m, Qm[ww * 32 + mt * 16 + (lane & 15)]);
            sq[mt] = m > 0.f ? m / 448.0f : 1.0f;
            const float rh = 1.0f / sq[mt], rl = rh * 16.0f;
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                float v[8], hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(__fmul_rn(bf2f(qf[mt][ks][e]), rh), -448.0f), 448.0f);
                qh[mt][ks] = pack_fp8x8(v);
                unpack_fp8x8(qh[mt][ks], hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float res = __fsub_rn(bf2f(qf[mt][ks][e]), __fmul_rn(hv[e], sq[mt]));
                    v[e] = fminf(fmaxf(__fmul_rn(res, rl), -448.0f), 448.0f);
                }
                ql[mt][ks] = pack_fp8x8(v);
            }
        }
    }
    // c2 = 2^ceil(log2(largest row scale of the chunk)): G = p x (scale / c2) x 8 <= 2^15
    int c2e;
    (void)frexpf(smax > 0.f ? smax : 1.0f, &c2e);         // smax = f x 2^c2e, f in [0.5, 1)  ->  smax <= 2^c2e
    const float g_mul = ldexpf(8.0f, -c2e);

    const int mt_o = w >> 1, nt_o = w & 1;
    const int q_o = q0 + mt_o * 16 + (lane & 15);
    const bool q_ok = q_o < a.B;
    const int64_t label = q_ok ? a.labels[q_o] - (int64_t)a.m_begin : -1;
    const float zs = sq[mt_o] * a.inv_tau;                // query scale x 1 / tau
    float st_m = -INFINITY, st_l = 0.f, st_sl = 0.f, st_lab = -INFINITY;
    f32x4 dq[2][NDT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[mt][dt] = f32x4{0, 0, 0, 0};
    __syncthreads();                                      // As (scale maxima) is reused below

    for (int t = 0; t < ntiles; ++t) {
        // tile t is in when at most the tiles requested after it are outstanding (its own successor in the ring is
        // requested only below: the dq phase of tile t - 1 read that buffer until the barrier)
        const int behind = min(S - 2, ntiles - 1 - t);
        if (behind >= 3) wait_vmcnt<3 * NDMA>();
        else if (behind == 2) wait_vmcnt<2 * NDMA>();
        else if (behind == 1) wait_vmcnt<NDMA>();
        else wait_vmcnt<0>();
        lds_barrier();
        if (t + S - 1 < ntiles) stage(t + S - 1);
        const char* T = tiles + (t % S) * RAW_B;
        f32x4 sh[2][2], sl4[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sh[mt][nt] = f32x4{0, 0, 0, 0};
                sl4[mt][nt] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int r = nt * 16 + (lane & 15);
                const int kb = w * DW + ks * 32 + (lane >> 4) * 8;               // byte offset of this lane's 8 values
                const long bfr = *(const long*)(T + r * D + (((kb >> 4) ^ fp8_swz<D>(r)) << 4) + (kb & 8));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    sh[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr, qh[mt][ks], sh[mt][nt], 0, 0, 0);
                    sl4[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr, ql[mt][ks], sl4[mt][nt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                *(f32x4*)(Sp + (((w * 2 + mt) * 2 + nt) * 64 + lane) * 4) = sh[mt][nt] + sl4[mt][nt] * 0.0625f;
        lds_barrier();
        // ---- statistics + G (bank_stream_kernel's FUSED branch: integer lazy reference exponent, label key kept out of G)
        f32x4 sv = {0, 0, 0, 0}, so = {0, 0, 0, 0};
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            sv += *(const f32x4*)(Sp + (((ww * 2 + mt_o) * 2 + nt_o) * 64 + lane) * 4);
            so += *(const f32x4*)(Sp + (((ww * 2 + mt_o) * 2 + (nt_o ^ 1)) * 64 + lane) * 4);
        }
        const int kl = t * TR + nt_o * 16 + (lane >> 4) * 4, klo = t * TR + (nt_o ^ 1) * 16 + (lane >> 4) * 4;   // chunk-local rows
        const int key0 = m_lo + kl, key0o = m_lo + klo;
        const f32x4 sb4 = *(const f32x4*)(Ssc + kl), sbo = *(const f32x4*)(Ssc + klo);
        float z[4], tmx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[r] = (key0 + r < a.M) ? sv[r] * sb4[r] * zs : -INFINITY;
            const float zo = (key0o + r < a.M) ? so[r] * sbo[r] * zs : -INFINITY;
            tmx = fmaxf(tmx, fmaxf(z[r], zo));
        }
        tmx = fmaxf(tmx, __shfl_xor(tmx, 16, 64));
        tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
        constexpr float LOG2E = 1.4426950408889634f;
        const float tm2 = tmx * LOG2E;
        const float mn = (tm2 > st_m + (float)FP8F_SLACK) ? ceilf(tm2) : st_m;
        const float alpha = (mn > st_m && st_m > -INFINITY) ? ldexpf(1.0f, (int)(st_m - mn)) : 1.0f;
        float add = 0.f;
        uint32_t g0 = 0, g1 = 0, g2 = 0;                  // this lane's 4 keys, one bf8 term each
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = z[r] > -INFINITY;
            const float z2 = z[r] * LOG2E, zf = floorf(z2);
            const float pv = live ? ldexpf(__builtin_amdgcn_exp2f(z2 - zf), (int)fmaxf(zf - mn, -200.0f)) : 0.f;
            add += pv;
            st_sl += live ? z[r] : 0.f;
            const bool is_lab = live && (int64_t)(key0 + r) == label;
            st_lab = is_lab ? z[r] : st_lab;
            x[r] = (q_ok && !is_lab) ? pv * (sb4[r] * g_mul) : 0.f;
        }
        st_l = st_l * alpha + add;
        st_m = mn;
        {
            // x = t0 + t1 + t2, every term the bf8 rounding of the remainder
            int p01 = __builtin_amdgcn_cvt_pk_bf8_f32(x[0], x[1], 0, false);
            p01 = __builtin_amdgcn_cvt_pk_bf8_f32(x[2], x[3], p01, true);
            g0 = (uint32_t)p01;
            f32x2 lo = __builtin_amdgcn_cvt_pk_f32_bf8(p01, false), hi = __builtin_amdgcn_cvt_pk_f32_bf8(p01, true);
            float r0 = x[0] - lo[0], r1 = x[1] - lo[1], r2 = x[2] - hi[0], r3 = x[3] - hi[1];
            int p1 = __builtin_amdgcn_cvt_pk_bf8_f32(r0, r1, 0, false);
            p1 = __builtin_amdgcn_cvt_pk_bf8_f32(r2, r3, p1, true);
            g1 = (uint32_t)p1;
            lo = __builtin_amdgcn_cvt_pk_f32_bf8(p1, false); hi = __builtin_amdgcn_cvt_pk_f32_bf8(p1, true);
            r0 -= lo[0]; r1 -= lo[1]; r2 -= hi[0]; r3 -= hi[1];
            if constexpr (GT == 3) {
                int p2 = __builtin_amdgcn_cvt_pk_bf8_f32(r0, r1, 0, false);
                p2 = __builtin_amdgcn_cvt_pk_bf8_f32(r2, r3, p2, true);
                g2 = (uint32_t)p2;
            }
        }
        {
            const int go = (mt_o * 16 + (lane & 15)) * TR + nt_o * 16 + (lane >> 4) * 4;
            *(uint32_t*)(G8 + go) = g0;
            *(uint32_t*)(G8 + BQ * TR + go) = g1;
            if constexpr (GT == 3) *(uint32_t*)(G8 + 2 * BQ * TR + go) = g2;
        }
        if (nt_o == 0 && lane < 16) As[mt_o * 16 + lane] = alpha;
        lds_barrier();
        {
            const float a0 = As[lane & 15], a1 = As[16 + (lane & 15)];
            if (__any(a0 != 1.0f || a1 != 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) { dq[0][dt] *= a0; dq[1][dt] *= a1; }
            }
        }
        // ---- dq[q][d] += sum_key G[q][key] raw[key][d]:  D[i = d][j = query], k = key (one 32-step, three bf8 terms)
        long gf[GT][2];
#pragma unroll
        for (int pl = 0; pl < GT; ++pl)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                gf[pl][mt] = *(const long*)(G8 + pl * BQ * TR + (mt * 16 + (lane & 15)) * TR + (lane >> 4) * 8);
        const int trow = (lane >> 4) * 8 + ((lane & 15) >> 1);                   // the row this lane hands to the transpose read
        const char* tbase = T + trow * D + (lane & 1) * 8;
        const int tsw = fp8_swz<D>(trow);
        long af[NDT];
        {
            v2i av[NDT];
#pragma unroll
            for (int pl = 0; pl < GT; ++pl) { lds_tie(gf[pl][0]); lds_tie(gf[pl][1]); }   // G fragments are in before the asm reads
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int c = (w * DW + dt * 16) >> 4;
                av[dt] = lds_tr8_b64_asm(tbase + ((c ^ tsw) << 4));      // asm read: no vmcnt(0) in front of it (common.h)
            }
            wait_lgkm<0>();
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                lds_tie(av[dt]);
                af[dt] = (long)(((unsigned long)(unsigned)av[dt][1] << 32) | (unsigned long)(unsigned)av[dt][0]);
            }
        }
        // term-major: the three products into one accumulator are 2 NDT MFMAs apart, never back to back
#pragma unroll
        for (int pl = 0; pl < GT; ++pl)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    dq[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(af[dt], gf[pl][mt], dq[mt][dt], 0, 0, 0);
    }

    // ---- chunk partial of dq (fp32, x c2 / 8) and the chunk statistics, as the FUSED epilogue of bank_stream_kernel
    const float o_mul = ldexpf(0.125f, c2e);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int q = q0 + mt * 16 + (lane & 15);
        if (q >= a.B) continue;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
            *(f32x4*)(Op + ((size_t)mi * a.B + q) * D + w * DW + dt * 16 + (lane >> 4) * 4) = dq[mt][dt] * o_mul;
    }
    __syncthreads();
    float* Fin = Sp;
    float* f = Fin + ((w * 64 + lane) * 4);
    f[0] = st_m; f[1] = st_l; f[2] = st_sl; f[3] = st_lab;
    __syncthreads();
    if (tid < BQ) {
        const int mt = tid >> 4, ql_ = tid & 15;
        float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
        for (int nt = 0; nt < 2; ++nt)
            for (int g = 0; g < 4; ++g) {
                const float* p = Fin + (((mt * 2 + nt) * 64 + g * 16 + ql_) * 4);
                m = fmaxf(m, p[0]);
                l += p[1];
                sl += p[2];
                lab = fmaxf(lab, p[3]);
            }
        const int q = q0 + tid;
        if (q < a.B) {
            float* o = sp + ((size_t)mi * a.B + q) * 4;
            o[0] = m * 0.6931471805599453f; o[1] = l; o[2] = sl; o[3] = lab;
        }
    }
    if (blockIdx.x == 0 && tid == 0) *(int*)(sp + (size_t)ck.nchunks * a.B * 4) = 0;   // bank_step_tail_kernel's arrival counter
}

// SPN_BANK_FP8_GTERMS=3: G of the all-fp8 fused pass as three e5m2 terms (round 3's form); default 2
static int fp8_gterms() {
    static const int n = [] {
        const char* e = spn_env("SPN_BANK_FP8_GTERMS");
        return (e && e[0] == '3') ? 3 : 2;
    }();
    return n;
}

template <int D, int GT>
static int launch_bank_fp8_fused_t(const BankArgs& a, const BankChunking& c, float* Op, float* sp, hipStream_t st) {
    const size_t lds = (size_t)fp8f_stages<D>() * TR * D + 4 * 4096 + 4 * BQ * 4 + FP8_MAX_CHUNK_ROWS * 4 + 3 * BQ * TR + BQ * 4;
    auto kern = bank_fp8_fused_kernel<D, GT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        const double bytes = (double)a.M * D + 4.0 * a.M + (double)a.B * D * 2 + (double)c.nchunks * a.B * (D * 4 + 16);
        ProfScope prof(PK_BANK_FWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(c.nq * c.nchunks), dim3(256), lds, st, a, c, Op, sp);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

template <int D>
static int launch_bank_fp8_fused(const BankArgs& a, const BankChunking& c, float* Op, float* sp, hipStream_t st) {
    return fp8_gterms() == 3 ? launch_bank_fp8_fused_t<D, 3>(a, c, Op, sp, st) : launch_bank_fp8_fused_t<D, 2>(a, c, Op, sp, st);
}

// SPN_BANK_FP8_MFMA=0 keeps the forward pass of an fp8 bank on the dequantise-to-bf16 path (A/B switch)
static bool bank_fp8_mfma_on() {
    static const bool on = [] {
        const char* e = spn_env("SPN_BANK_FP8_MFMA");
        return !(e && e[0] == '0');
    }();
    return on;
}

// ------------------------------------------------------------------ token-max bank, wave-private tiles
// Same contract as bank_stream_kernel<.., GRP = true>, restructured for small D (the Q-Former's 256): the
// block kernel above splits D over its four waves and pays three block barriers per 16 KB tile.  Here every wave owns
// whole tiles (= targets): it DMAs its own tile into its own double-buffered LDS region, computes the 32 x 32 logits
// over the full D, takes the max / arg-max over the 32 token rows with lane shuffles, and (backward) builds the
// one-non-zero-per-query G operand directly in registers - no block barrier inside the main loop, the only waits are
// the wave's own counted vmcnt.  Waves of a block take tiles w, w+4, ... of the block's chunk; their statistics /
// dq partials are merged through LDS once at the end.
template <int D, bool BWD>
__global__ __launch_bounds__(256, 1) void bank_tokmax_kernel(BankArgs a, BankChunking ck, const float* __restrict__ row_lse,
                                                            float label_smoothing, float inv_m_total,
                                                            float* __restrict__ ws) {
    constexpr int ROWB = D * 2, TILE_B = TR * ROWB;
    constexpr int KS = D / 32, NDT = D / 16, NDMA = TILE_B / 1024;
    static_assert(2 * TILE_B == BQ * D * 4, "a wave's two tile buffers hold its fp32 dq partial");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    char* my = smem + w * 2 * TILE_B;
    float* Fin = (float*)(smem + 8 * TILE_B);             // [4 waves][BQ][4]
    const int mi = blockIdx.x % ck.nchunks, qi = blockIdx.x / ck.nchunks;
    const int q0 = qi * BQ;
    const int m_lo = mi * ck.rows;
    const int m_hi = min(a.M, m_lo + ck.rows);
    const int ntiles = m_hi > m_lo ? (m_hi - m_lo) / TR : 0;      // M % TR == 0 (checked by the host)
    const int nmine = ntiles > w ? (ntiles - w + 3) / 4 : 0;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bank, (uint32_t)a.M * (uint32_t)ROWB);

    bf16x8 qf[2][KS];
    int64_t label[2];
    float lse[2];
    bool q_ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int qr = q0 + mt * 16 + (lane & 15);
        q_ok[mt] = qr < a.B;
        label[mt] = q_ok[mt] ? a.labels[qr] - (int64_t)a.m_begin : -1;
        lse[mt] = (BWD && q_ok[mt]) ? row_lse[qr] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (q_ok[mt]) {
                qf[mt][ks] = *(const bf16x8*)(a.q + (size_t)qr * a.ldq + ks * 32 + (lane >> 4) * 8);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[mt][ks][e] = (bf16_t)0.0f;
            }
        }
    }
    float st_m[2] = {-INFINITY, -INFINITY}, st_l[2] = {0.f, 0.f}, st_sl[2] = {0.f, 0.f}, st_lab[2] = {-INFINITY, -INFINITY};
    [[maybe_unused]] f32x4 dq[2][NDT];
    if constexpr (BWD) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[mt][dt] = f32x4{0, 0, 0, 0};
    }
    auto stage = [&](int t, int buf) {
        char* dst = my + buf * TILE_B;
        const int mrow0 = m_lo + t * TR;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int p = i * 1024 + lane * 16;
            const int r = p / ROWB, cp = (p % ROWB) >> 4;
            const int c = cp ^ bank_swz(r & 15);
            glds16(rs, dst + i * 1024, (uint32_t)(mrow0 + r) * (uint32_t)ROWB + (uint32_t)c * 16u);
        }
    };
    if (nmine > 0) stage(w, 0);
    for (int i = 0; i < nmine; ++i) {
        const int t = w + 4 * i, buf = i & 1;
        if (i + 1 < nmine) {
            stage(t + 4, buf ^ 1);                 // the other buffer: last read one iteration ago, by this wave only
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        } else {
            wait_vm0();
        }
        const char* T = my + buf * TILE_B;
        f32x4 s[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) s[mt][nt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int r = nt * 16 + (lane & 15);
                const int c = ks * 4 + (lane >> 4);
                const bf16x8 bfrag = *(const bf16x8*)(T + r * ROWB + ((c ^ bank_swz(r & 15)) << 4));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) s[mt][nt] = mfma16b(bfrag, qf[mt][ks], s[mt][nt]);
            }
        }
        const int tgt = m_lo / TR + t;             // shard-local target id of this tile
        [[maybe_unused]] bf16x8 gf[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // lane holds keys nt*16 + (lane>>4)*4 + r of query lane&15: max + FIRST arg-max over the 32 keys
            float gm = -INFINITY;
            int gi = TR;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = nt * 16 + (lane >> 4) * 4 + r;
                    const float v = s[mt][nt][r];
                    if (v > gm || (v == gm && k < gi)) { gm = v; gi = k; }
                }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const float m2 = __shfl_xor(gm, off, 64);
                const int i2 = __shfl_xor(gi, off, 64);
                if (m2 > gm || (m2 == gm && i2 < gi)) { gm = m2; gi = i2; }
            }
            const float z = gm * a.inv_tau;
            if constexpr (!BWD) {
                const float mn = fmaxf(st_m[mt], z);
                st_l[mt] = st_l[mt] * __expf(st_m[mt] - mn) + __expf(z - mn);
                st_m[mt] = mn;
                st_sl[mt] += z;
                if ((int64_t)tgt == label[mt]) st_lab[mt] = z;
            } else {
                float gv = 0.f;
                if (q_ok[mt]) {
                    gv = __expf(z - lse[mt]) - label_smoothing * inv_m_total;
                    if ((int64_t)tgt == label[mt]) gv -= 1.0f - label_smoothing;
                }
                // B operand of the dq MFMA: j = query (lane&15), k = key (lane>>4)*8 + e
                const bf16_t gvb = f2bf(gv), zb = (bf16_t)0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) gf[mt][e] = ((lane >> 4) * 8 + e == gi) ? gvb : zb;
            }
        }
        if constexpr (BWD) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                union { s16x4 h[2]; bf16x8 v; } u;
                const int cb = dt * 16;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int r = (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
                    const int col = cb + (lane & 3) * 4;
                    u.h[h] = lds_tr16_b64(T + r * ROWB + (((col >> 3) ^ bank_swz(r & 15)) << 4) + (col & 7) * 2);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) dq[mt][dt] = mfma16b(u.v, gf[mt], dq[mt][dt]);
            }
        }
    }
    if constexpr (!BWD) {
        if (lane < 16) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                *(f32x4*)(Fin + ((w * BQ + mt * 16 + lane) * 4)) = f32x4{st_m[mt], st_l[mt], st_sl[mt], st_lab[mt]};
        }
        __syncthreads();
        if (tid < BQ) {
            float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
            for (int ww = 0; ww < 4; ++ww) {
                const f32x4 p = *(const f32x4*)(Fin + ((ww * BQ + tid) * 4));
                const float mn = fmaxf(m, p[0]);
                if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
                m = mn;
                sl += p[2];
                lab = fmaxf(lab, p[3]);
            }
            const int q = q0 + tid;
            if (q < a.B) *(f32x4*)(ws + ((size_t)mi * a.B + q) * 4) = f32x4{m, l, sl, lab};
        }
    } else {
        // this wave's dq partial -> its own (now idle) tile buffers as fp32 [BQ][D]; then all threads add the 4 partials
        float* mine = (float*)my;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
                *(f32x4*)(mine + (mt * 16 + (lane & 15)) * D + dt * 16 + (lane >> 4) * 4) = dq[mt][dt];
        __syncthreads();
        for (int e = tid * 4; e < BQ * D; e += 256 * 4) {
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) acc += *(const f32x4*)((const float*)(smem + ww * 2 * TILE_B) + e);
            const int q = q0 + e / D;
            if (q < a.B) *(f32x4*)(ws + ((size_t)mi * a.B + q) * D + e % D) = acc;
        }
    }
}

template <int D, bool BWD>
static int launch_tokmax(const BankArgs& a, const BankChunking& c, const float* row_lse, float ls, float inv_m,
                         float* ws, hipStream_t st) {
    const size_t lds = 8 * (size_t)TR * D * 2 + 4 * BQ * 4 * sizeof(float);
    auto kern = bank_tokmax_kernel<D, BWD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        const double bytes = (double)a.M * D * 2 + (double)a.B * D * (BWD ? 6 : 2) + (double)a.B * 16;
        ProfScope prof(BWD ? PK_BANK_BWD : PK_BANK_FWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(c.nq * c.nchunks), dim3(256), lds, st, a, c, row_lse, ls, inv_m, ws);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// SPN_TOKMAX_BLOCK=1 keeps the block-cooperative kernel for D <= 256 too (A/B switch for the tests)
static bool tokmax_wave_path(const BankArgs& a) {
    static const bool off = [] {
        const char* e = spn_env("SPN_TOKMAX_BLOCK");
        return e && e[0] == '1';
    }();
    return a.group && !off && (a.D == 128 || a.D == 256);
}

// fold per-chunk statistics [n][B][4] -> [B][4]
// one wave per query row: lanes take chunks i = lane, lane+64, ... then merge across the wave
__global__ void bank_stats_fold_kernel(const float* __restrict__ ws, int n, int B, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= B) return;
    float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        const f32x4 p = *(const f32x4*)(ws + ((size_t)i * B + q) * 4);
        const float mn = fmaxf(m, p[0]);
        if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
        m = mn;
        sl += p[2];
        lab = fmaxf(lab, p[3]);
    }
    const float mw = wave_max(m);
    l = (m > -INFINITY) ? l * __expf(m - mw) : 0.f;
    l = wave_sum(l);
    sl = wave_sum(sl);
    lab = wave_max(lab);
    if (lane == 0) {
        float* o = stats + (size_t)q * 4;
        o[0] = mw; o[1] = l; o[2] = sl; o[3] = lab;
    }
}

int bank_stats_fold(const float* ws, int n, int B, float* stats, hipStream_t st) {
    hipLaunchKernelGGL(bank_stats_fold_kernel, dim3((B + 3) / 4), dim3(256), 0, st, ws, n, B, stats);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// loss_row = lse - (1-eps)*label_logit - eps*mean_j logit_j   (CrossEntropyLoss w/ label smoothing)
__global__ void bank_loss_finalize_kernel(const float* __restrict__ stats, int nshards, int B, float inv_m_total,
                                          float label_smoothing, float* __restrict__ row_lse,
                                          float* __restrict__ row_loss, float* __restrict__ loss_mean) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int q = threadIdx.x; q < B; q += blockDim.x) {
        float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
        for (int i = 0; i < nshards; ++i) {
            const float* p = stats + ((size_t)i * B + q) * 4;
            const float mn = fmaxf(m, p[0]);
            if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
            m = mn;
            sl += p[2];
            lab = fmaxf(lab, p[3]);
        }
        const float lse = m + logf(l);
        const float loss = lse - (1.0f - label_smoothing) * lab - label_smoothing * sl * inv_m_total;
        if (row_lse) row_lse[q] = lse;
        if (row_loss) row_loss[q] = loss;
        acc += loss;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_mean) *loss_mean = red[0] / (float)B;
}

size_t bank_workspace_bytes(int B, int M, int D) {
    const BankChunking c = bank_chunking(B, M);
    size_t a = (size_t)c.nchunks * B * 4 * sizeof(float);
    const size_t g = (size_t)((M + 127) / 128) * B * 4 * sizeof(float);   // GEMM-path statistics partials (the finer of the two tilings)
    if (g > a) a = g;
    const size_t b = (size_t)c.nchunks * B * D * sizeof(float);
    if (b > a) a = b;
    const size_t n = bank2_workspace_bytes(B, M, D);                              // second-generation kernels (bank2.hip)
    if (n > a) a = n;
    const size_t t = B >= 128 && B % 8 == 0 ? gemm_tn_workspace_bytes(M, B, D) : 0;  // GEMM backward pass from saved p
    return a > t ? a : t;
}

template <int D, bool BWD, bool FP8, bool GRP = false>
static int launch_bank(const BankArgs& a, const BankChunking& c, const float* row_lse, float ls, float inv_m,
                       float* ws, hipStream_t st) {
    const size_t lds = bank_stream_tiles_bytes<D, FP8>() + 4 * 4096 + (size_t)BQ * (TR + 8) * 2;
    auto kern = bank_stream_kernel<D, BWD, FP8, GRP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        // "work" = algorithmic HBM bytes of one pass: the bank shard once + q in (+ dq out)
        const double bytes = (double)a.M * D * (FP8 ? 1 : 2) + (FP8 ? 4.0 * a.M : 0.0) + (double)a.B * D * (BWD ? 6 : 2) +
                             (double)a.B * 16;
        ProfScope prof(BWD ? PK_BANK_BWD : PK_BANK_FWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(c.nq * c.nchunks), dim3(256), lds, st, a, c, row_lse, ls, inv_m, ws, (float*)nullptr);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ---- fused forward + backward pass (one bank read per step); save buffer = [nchunks][B][D] O partials, [nchunks][B][4] stats
// Routing of the forward / backward pair, process-wide (spn_bank_config): 0 = default, 1 = second-generation streaming pair
// below 128 queries (csrc/bank2.hip - compiled only into -DSPN_EXPERIMENTS builds: measured slower, LABNOTES.md section 5.4),
// 2 = the fused single pass also at B >= 256, 3 = two passes everywhere, 4 = default routing with the e4m3 fused pass on the
// kernel that keeps a bf16 tile image.  Environment defaults (read when the library is loaded): SPN_BANK2=1 -> 1,
// SPN_BANK_FUSED_LARGE=1 -> 2, SPN_BANK_FUSED=0 -> 3.
static int g_bank_mode = -1;
int bank_config(int mode) {
#ifndef SPN_EXPERIMENTS
    if (mode == 1) return SPN_ERR_ARG;          // the streaming pair is not part of this build
#endif
    g_bank_mode = (mode >= 0 && mode <= 4) ? mode : 0;
    return SPN_OK;
}
int bank_mode() {
    if (g_bank_mode < 0) {
        auto is = [](const char* n, char c) { const char* e = spn_env(n); return e && e[0] == c; };
        g_bank_mode = is("SPN_BANK_FUSED_LARGE", '1') ? 2 : is("SPN_BANK_FUSED", '0') ? 3 : 0;
#ifdef SPN_EXPERIMENTS
        if (is("SPN_BANK2", '1')) g_bank_mode = 1;
#endif
    }
    return g_bank_mode;
}
int bank_saved_ld(int M) { return (M + 31) / 32 * 32; }
size_t bank_saved_bytes(int B, int M) { return (size_t)B * bank_saved_ld(M) * sizeof(float); }
#ifndef SPN_EXPERIMENTS
bool bank_saved_path(const BankArgs&) { return false; }
size_t bank2_workspace_bytes(int, int, int) { return 0; }
int bank2_stats_fwd(const BankArgs&, float*, float*, float*, size_t, hipStream_t) { return SPN_ERR_ARG; }
int bank2_grad_q(const BankArgs&, const float*, const float*, float, int64_t, float, float*, float*, size_t, hipStream_t) {
    return SPN_ERR_ARG;
}
#endif

static bool bank_fused_on() { return bank_mode() != 3; }
// mode 2: the fused stream pass also at B >= 256, D >= 512, where the default is the GEMM forward pass that keeps p + the
// G^T / TN-GEMM backward pass
static bool bank_fused_large() { return bank_mode() == 2; }
bool bank_fused_ok(const BankArgs& a) {
    if (!bank_fused_on() || a.group || a.B <= 0) return false;
    if (a.bank_scale && a.B >= 256) return false;          // large e4m3 batches expand the shard once per pass instead
    switch (a.D) {
        case 128: case 256: case 512: case 640: case 768: case 1024: return true;
        default: return false;
    }
}
// workgroups of the fused pass: 512 (two per CU) for an e4m3 bank once a workgroup of a 256-block grid would walk 24 or more
// tiles (measured: equal at 100 000 rows and B = 32, 141 -> 103 us at 400 000 rows; at 40 000 rows the extra partials cost
// more than the overlap gives), 256 otherwise
static int fused_blocks(int B, int M, bool fp8, int D) {
    const int nq = (B + BQ - 1) / BQ, per = ((M + TR - 1) / TR) / (256 / nq > 0 ? 256 / nq : 1);
    return (fp8 && D <= 768 && per >= 24) ? 512 : 256;
}
static size_t fused_save_bytes(int B, int M, int D) {
    const BankChunking c = bank_chunking(B, M, fused_blocks(B, M, true, 768));     // the larger of the two chunk counts
    return (size_t)c.nchunks * B * (D + 4) * sizeof(float) + 64;     // fp32 partials + {m, l, sum z, label z} + arrival counter
}

template <int D, bool FP8>
static int launch_bank_fused(const BankArgs& a, const BankChunking& c, float* Op, float* sp, hipStream_t st) {
    const size_t lds = bank_stream_tiles_bytes<D, FP8>() + 4 * 4096 + (size_t)BQ * (TR + 8) * 2 + BQ * sizeof(float) +
                       (FP8 ? (4 * BQ + 2 * TR) * sizeof(float) : 0);       // + Qm, Sst
    auto kern = bank_stream_kernel<D, true, FP8, false, true>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        // ONE read of the shard (the step needs no second pass) + the chunk partials written for the fold
        const double bytes = (double)a.M * D * (FP8 ? 1 : 2) + (FP8 ? 4.0 * a.M : 0.0) + (double)a.B * D * 2 +
                             (double)c.nchunks * a.B * (D * 4 + 16);
        ProfScope prof(PK_BANK_FWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(c.nq * c.nchunks), dim3(256), lds, st, a, c, (const float*)nullptr, 0.f, 0.f, Op, sp);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

static int bank_fused_fwd(const BankArgs& a, float* stats, float* save, hipStream_t st) {
    const BankChunking c = bank_chunking(a.B, a.M, fused_blocks(a.B, a.M, a.bank_scale != nullptr, a.D));
    float* Op = save;                                                    // [nchunks][B][D]
    float* sp = save + (size_t)c.nchunks * a.B * a.D;
    int rc = SPN_ERR_SHAPE;
    // e4m3 bank: the all-fp8-MFMA kernel when the chunk's row scales fit its LDS table (SPN_BANK_FP8_FUSED=0: the kernel that
    // dequantises each tile into a bf16 image for the dq GEMM)
    static const bool f8 = [] {
        const char* e = spn_env("SPN_BANK_FP8_FUSED");
        return !(e && e[0] == '0');
    }();
    const bool f8k = f8 && bank_mode() != 4 && a.bank_scale && c.rows <= FP8_MAX_CHUNK_ROWS;
#define SPN_FUSED(D_) case D_: rc = f8k ? launch_bank_fp8_fused<D_>(a, c, Op, sp, st) : a.bank_scale ? launch_bank_fused<D_, true>(a, c, Op, sp, st) : launch_bank_fused<D_, false>(a, c, Op, sp, st); break;
    switch (a.D) {
        SPN_FUSED(128) SPN_FUSED(256) SPN_FUSED(512) SPN_FUSED(640) SPN_FUSED(768) SPN_FUSED(1024)
        default: return SPN_ERR_SHAPE;
    }
#undef SPN_FUSED
    if (rc) return rc;
    return bank_stats_fold(sp, c.nchunks, a.B, stats, st);
}

static int bank_fused_bwd(const BankArgs& a, const float* save, const float* row_lse, float grad_scale, float* dq, hipStream_t st) {
    const BankChunking c = bank_chunking(a.B, a.M, fused_blocks(a.B, a.M, a.bank_scale != nullptr, a.D));
    const float* Op = save;
    const float* sp = save + (size_t)c.nchunks * a.B * a.D;
    ProfScope prof(PK_BANK_BWD, (double)c.nchunks * a.B * (a.D * 4 + 16) + (double)a.B * a.D * 6, st);
    const dim3 grid((a.D + 63) / 64, a.B);
    if (a.bank_scale)
        hipLaunchKernelGGL(bank_fused_combine_kernel<true>, grid, dim3(256), 0, st, Op, sp, c.nchunks, a.B, a.D, row_lse, (const void*)a.bank,
                           a.bank_scale, a.labels, a.m_begin, a.M, grad_scale * a.inv_tau, dq, a.D);
    else
        hipLaunchKernelGGL(bank_fused_combine_kernel<false>, grid, dim3(256), 0, st, Op, sp, c.nchunks, a.B, a.D, row_lse,
                           (const void*)a.bank, (const float*)nullptr, a.labels, a.m_begin, a.M, grad_scale * a.inv_tau, dq, a.D);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// One shard holding the whole bank, no label smoothing: the step's forward AND backward w.r.t. the queries in two launches
// (the single pass over the bank + bank_step_tail_kernel) instead of four (pass, statistics fold, finalize, combine).
static int bank_check(const BankArgs& a);
// 192..256 queries (the GEMM-shaped pair): the five launches of the saved-probability path behind ONE call - statistics pass (keeps
// p^T), fold + finalize + mean in one tail launch, G^T scaling, dq GEMM, split-k fold.  Scratch behind the LargeSave image of `save`:
// the statistics partials, the tail's arrival counter, the dq GEMM's split-k slabs (sized for the widest bank: the C-ABI sizes the
// scratch from (B, M) alone).
static bool bank_s160_on();
static bool bank_step_large_ok(const BankArgs& a) {
    return bank_saved_path_large(a) && !bank_fused_large() && bank_s160_on() && bank_stats160_ok(a.B, a.D, a.ldq) && a.D <= 1024 &&
           (uint64_t)a.M * a.D * 2 < (1ull << 32);
}
static size_t large_save_bytes(int B, int M);
static size_t large_step_extra_bytes(int B, int M) {
    size_t tn = 0;                                     // the split count depends on the width: take the largest need
    for (int D = 512; D <= 1024; D += 64) {
        const size_t t = gemm_tn_workspace_bytes(M, B, D);
        if (t > tn) tn = t;
    }
    return (((size_t)bank_stats160_tiles(M) * B * 16 + 255) & ~(size_t)255) + 256 + tn;
}
static int bank_step_large(const BankArgs& a, float* save, float grad_scale, float* row_lse, float* row_loss, float* loss_mean,
                           float* dq, hipStream_t st);
bool bank_step_ok(const BankArgs& a) {
    if (bank_step_large_ok(a)) return true;
    return bank_fused_ok(a) && !bank_saved_path(a) && !(bank_saved_path_large(a) && !bank_fused_large());
}
int bank_step(const BankArgs& a, float* save, float grad_scale, float* row_lse, float* row_loss, float* loss_mean, float* dq,
              hipStream_t st) {
    int rc = bank_check(a);
    if (rc) return rc;
    if (!save || !row_lse || !row_loss || !loss_mean || !dq || a.m_begin != 0) return SPN_ERR_ARG;
    if (!bank_step_ok(a)) return SPN_ERR_SHAPE;
    if (bank_step_large_ok(a)) return bank_step_large(a, save, grad_scale, row_lse, row_loss, loss_mean, dq, st);
    const BankChunking c = bank_chunking(a.B, a.M, fused_blocks(a.B, a.M, a.bank_scale != nullptr, a.D));
    float* Op = save;
    float* sp = save + (size_t)c.nchunks * a.B * a.D;
    static const bool f8 = [] {
        const char* e = spn_env("SPN_BANK_FP8_FUSED");
        return !(e && e[0] == '0');
    }();
    const bool f8k = f8 && bank_mode() != 4 && a.bank_scale && c.rows <= FP8_MAX_CHUNK_ROWS;
    rc = SPN_ERR_SHAPE;
#define SPN_FUSED(D_) case D_: rc = f8k ? launch_bank_fp8_fused<D_>(a, c, Op, sp, st) : a.bank_scale ? launch_bank_fused<D_, true>(a, c, Op, sp, st) : launch_bank_fused<D_, false>(a, c, Op, sp, st); break;
    switch (a.D) {
        SPN_FUSED(128) SPN_FUSED(256) SPN_FUSED(512) SPN_FUSED(640) SPN_FUSED(768) SPN_FUSED(1024)
        default: return SPN_ERR_SHAPE;
    }
#undef SPN_FUSED
    if (rc) return rc;
    ProfScope prof(PK_BANK_BWD, (double)c.nchunks * a.B * (a.D * 4 + 16) + (double)a.B * a.D * 6, st);
    const dim3 grid((a.D + 63) / 64, a.B);
    int* counter = (int*)(sp + (size_t)c.nchunks * a.B * 4);
    if (a.bank_scale)
        hipLaunchKernelGGL(bank_step_tail_kernel<true>, grid, dim3(256), 0, st, Op, sp, c.nchunks, a.B, a.D, (const void*)a.bank,
                           a.bank_scale, a.labels, a.M, grad_scale * a.inv_tau, dq, a.D, row_lse, row_loss, loss_mean, counter);
    else
        hipLaunchKernelGGL(bank_step_tail_kernel<false>, grid, dim3(256), 0, st, Op, sp, c.nchunks, a.B, a.D, (const void*)a.bank,
                           (const float*)nullptr, a.labels, a.M, grad_scale * a.inv_tau, dq, a.D, row_lse, row_loss, loss_mean, counter);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

#define SPN_BANK_CASE(D_, BWD_, ...)                                                                  \
    case D_:                                                                                          \
        rc = a.group ? launch_bank<D_, BWD_, false, true>(__VA_ARGS__)                                   \
             : a.bank_scale ? launch_bank<D_, BWD_, true>(__VA_ARGS__) : launch_bank<D_, BWD_, false>(__VA_ARGS__); \
        break;
#define SPN_BANK_DISPATCH(BWD_, ...)               \
    switch (a.D) {                                 \
        SPN_BANK_CASE(128, BWD_, __VA_ARGS__)      \
        SPN_BANK_CASE(256, BWD_, __VA_ARGS__)      \
        SPN_BANK_CASE(512, BWD_, __VA_ARGS__)      \
        SPN_BANK_CASE(640, BWD_, __VA_ARGS__)      \
        SPN_BANK_CASE(768, BWD_, __VA_ARGS__)      \
        SPN_BANK_CASE(1024, BWD_, __VA_ARGS__)     \
        default: return SPN_ERR_SHAPE;             \
    }

static int bank_check(const BankArgs& a) {
    if (a.B <= 0 || a.M <= 0 || !a.q || !a.bank || !a.labels) return SPN_ERR_ARG;
    if (a.ldq % 8 || a.ldq < a.D) return SPN_ERR_SHAPE;
    if ((uint64_t)a.M * a.D * (a.bank_scale ? 1 : 2) >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (a.group && (a.group != TR || a.M % TR || a.bank_scale)) return SPN_ERR_SHAPE;   // token-max: 32 bf16 rows per target
    return SPN_OK;
}

// e4m3 bytes [M, D] x per-row scale -> bf16 [M, D]: at large batches (every bank tile is used by many query tiles) the
// fp8 bank is expanded ONCE per pass into scratch and the bf16 kernels run on it - the in-kernel dequantisation of the
// streaming path repeats per query tile (B = 256: 96 -> ~57 us forward, 115 -> ~88 us backward).
__global__ void bank_dequant_fp8_kernel(const uint8_t* __restrict__ src, const float* __restrict__ scale,
                                        bf16_t* __restrict__ dst, size_t n16, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 16;
        const float sc = scale[e / D];
        const u32x4 v = *(const u32x4*)(src + e);
        bf16x8 o[2];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[d], false);
            const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[d], true);
            o[d >> 1][(d & 1) * 4 + 0] = f2bf(lo[0] * sc);
            o[d >> 1][(d & 1) * 4 + 1] = f2bf(lo[1] * sc);
            o[d >> 1][(d & 1) * 4 + 2] = f2bf(hi[0] * sc);
            o[d >> 1][(d & 1) * 4 + 3] = f2bf(hi[1] * sc);
        }
        *(bf16x8*)(dst + e) = o[0];
        *(bf16x8*)(dst + e + 8) = o[1];
    }
}

// the bf16 image of an e4m3 bank (exactly what the kernels' per-pass expansion holds): for callers that keep it across steps
int bank_dequant_fp8(const uint8_t* data, const float* scale, int M, int D, bf16_t* out, hipStream_t st) {
    if (!data || !scale || !out || M <= 0 || D <= 0 || D % 16) return SPN_ERR_ARG;
    const size_t n16 = (size_t)M * D / 16;
    const int blocks = (int)((n16 + 255) / 256 > 4096 ? 4096 : (n16 + 255) / 256);
    hipLaunchKernelGGL(bank_dequant_fp8_kernel, dim3(blocks), dim3(256), 0, st, data, scale, out, n16, D);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

static constexpr int FP8_EXPAND_MIN_B = 128;

size_t bank_workspace_bytes_fp8(int B, int M, int D) {
    size_t b = (bank_workspace_bytes(B, M, D) + 255) & ~(size_t)255;
    if (B >= FP8_EXPAND_MIN_B) b += (size_t)M * D * 2;
    return b;
}

// -> true (and *out filled) when the call should run on an expanded bf16 copy placed behind the regular workspace
static bool bank_expand_fp8(const BankArgs& a, float* ws, size_t ws_bytes, BankArgs* out, size_t* base, hipStream_t st) {
    if (!a.bank_scale || a.group || a.B < FP8_EXPAND_MIN_B || a.D % 16) return false;
    *base = (bank_workspace_bytes(a.B, a.M, a.D) + 255) & ~(size_t)255;
    if (ws_bytes < *base + (size_t)a.M * a.D * 2) return false;
    bf16_t* deq = (bf16_t*)((char*)ws + *base);
    const size_t n16 = (size_t)a.M * a.D / 16;
    const int blocks = (int)((n16 + 255) / 256 > 4096 ? 4096 : (n16 + 255) / 256);
    hipLaunchKernelGGL(bank_dequant_fp8_kernel, dim3(blocks), dim3(256), 0, st, (const uint8_t*)a.bank, a.bank_scale, deq, n16,
                       a.D);
    *out = a;
    out->bank = deq;
    out->bank_scale = nullptr;
    return true;
}

// ---------------------------------------------------------------------- batches of 128 queries and more, saved pass
// The forward pass is a GEMM (gemm_bank_stats); with a save buffer its epilogue also keeps p = exp(logit - tile max) as
// bf16 [B, ldp] and the tile maxima.  The backward pass is then two launches instead of a recomputing stream kernel:
//   G^T[m][b] = p * exp(tile max - lse_b) - [m == label_b] (1 - eps) - eps / M_total     (bank_p_to_gt_kernel, a transpose)
//   dq [B, D] = (G^T)^T bank = gemm_tn(G^T [M, B], bank [M, D])                          (the weight-gradient GEMM)
// i.e. 2 M B D flop on the MFMA GEMM kernels and 2 x 2 M B bytes of p / G traffic, against a recomputation of the logits
// plus a dq GEMM in 32-query blocks that re-read every bank tile B / 32 times from L2 (72 us at B = 256, M = 40 000).
struct LargeSave {
    bf16_t* P; int ldp; bf16_t* Gt; int mpad; float* tmax; int nt;
};
static size_t large_save_bytes(int B, int M) {
    const size_t ldp = (size_t)(M + 255) / 256 * 256, mpad = (size_t)(M + 63) / 64 * 64, nt = (size_t)(M + 127) / 128;   // the finer tiling
    return (((size_t)B * ldp * 2 + 255) & ~(size_t)255) + ((mpad * B * 2 + 255) & ~(size_t)255) + nt * B * 4;
}
static LargeSave large_save_at(void* base, int B, int M) {
    LargeSave s;
    s.ldp = (M + 255) / 256 * 256; s.mpad = (M + 63) / 64 * 64; s.nt = (M + 127) / 128;
    char* p = (char*)base;
    s.P = (bf16_t*)p; p += ((size_t)B * s.ldp * 2 + 255) & ~(size_t)255;
    s.Gt = (bf16_t*)p; p += ((size_t)s.mpad * B * 2 + 255) & ~(size_t)255;
    s.tmax = (float*)p;
    return s;
}
// SPN_BANK_S160=0: 128..256 queries stay on the 256 x 256-tile statistics GEMM (A/B switch)
static bool bank_s160_on() {
    static const bool on = [] {
        const char* e = spn_env("SPN_BANK_S160");
        return !(e && e[0] == '0');
    }();
    return on;
}
static bool bank_gemm_on() {
    static const bool use_gemm = [] {
        const char* e = spn_env("SPN_BANK_GEMM");
        return !(e && e[0] == '0');
    }();
    return use_gemm;
}
// batches that take the GEMM forward pass and keep p for a GEMM backward pass.  Measured (profiles/r03_bank_bench.txt):
// B = 256, D = 768, M = 40 000: backward 72 -> 60 us; B = 128, D = 256, M = 30 000 (the BLIP head): 22 -> 50 us - a
// 128 x 256 output is two tiles, the split-K GEMM cannot fill the chip with it - so the path is taken from 256 queries and
// 512 columns; below that the recomputing stream kernel stays.
bool bank_saved_path_large(const BankArgs& a) {
    // crossover against the fused stream pass at M = 40 000, D = 768 (rocprofv3, pair vs fused): B = 160: 81.7 vs 77.7 us, 192: 83.3
    // vs 87.4, 224: 88.7 vs 96.5, 256: 90.7 vs 111
    static const int min_b = env_int_min1("SPN_BANK_GEMM_PAIR_MIN", 192);
    return bank_gemm_on() && !a.bank_scale && !a.group && a.B >= min_b && a.B % 8 == 0 && a.D >= 512 && a.D % 64 == 0;
}
static size_t fused_save_bytes(int B, int M, int D);
static int bank_step_large(const BankArgs& a, float* save, float grad_scale, float* row_lse, float* row_loss, float* loss_mean,
                           float* dq, hipStream_t st) {
    const LargeSave sv = large_save_at(save, a.B, a.M);
    char* p = (char*)save + large_save_bytes(a.B, a.M);
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    float* partial = (float*)p;
    const int nt160 = bank_stats160_tiles(a.M);
    p += ((size_t)nt160 * a.B * 16 + 255) & ~(size_t)255;
    int* counter = (int*)p;
    p += 256;
    float* ws = (float*)p;
    const size_t ws_bytes = gemm_tn_workspace_bytes(a.M, a.B, a.D);
    int rc;
    {
        ProfScope prof(PK_BANK_FWD, (double)a.M * a.D * 2 + (double)a.B * a.D * 2 + (double)a.B * 16, st);
        rc = bank_stats160(a.q, a.ldq, a.bank, a.labels, a.B, a.M, a.D, 0, a.inv_tau, partial, sv.P, sv.tmax, st, counter);
        if (rc) return rc;
        rc = bank_stats_tail(partial, nt160, a.B, row_lse, row_loss, loss_mean, counter, st);
        if (rc) return rc;
    }
    ProfScope prof(PK_BANK_BWD, (double)a.M * a.D * 2 + (double)a.B * a.D * 6 + (double)a.B * 16, st);
    rc = bank_gt_scale(sv.P, sv.Gt, sv.tmax, row_lse, a.labels, a.B, a.M, 0, 0.f, 1.0f / (float)a.M, st);
    if (rc) return rc;
    return gemm_tn(sv.Gt, a.bank, a.M, a.B, a.D, a.B, a.D, dq, a.D, grad_scale * a.inv_tau, 0, nullptr, ws, ws_bytes, st);
}

size_t bank_saved_bytes_any(int B, int M) {
    const size_t small = bank_saved_bytes(B, M);
    size_t large = B >= 128 ? large_save_bytes(B, M) : 0;
    if (B >= 128 && B <= 256) large += 256 + large_step_extra_bytes(B, M);        // bank_step_large's scratch behind the image
    const size_t fused = fused_save_bytes(B, M, 1024);      // the widest bank: the C-ABI sizes the scratch from (B, M) alone
    const size_t a = small > large ? small : large;
    return a > fused ? a : fused;
}

__global__ __launch_bounds__(256) void bank_p_to_gt_kernel(const bf16_t* __restrict__ P, int ldp, const float* __restrict__ tmax,
                                                          const float* __restrict__ lse, const int64_t* __restrict__ labels,
                                                          int B, int M, int m_begin, float ls, float inv_m,
                                                          bf16_t* __restrict__ Gt, int tile_shift) {
    __shared__ float T[64][65];
    const int m0 = blockIdx.x * 64, b0 = blockIdx.y * 64, t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int bl = (t >> 3) + 32 * i, b = b0 + bl, m8 = (t & 7) * 8;
        float g[8];
        if (b < B) {
            const bf16x8 p = *(const bf16x8*)(P + (size_t)b * ldp + m0 + m8);
            const float tm = tmax[(size_t)((m0 + m8) >> tile_shift) * B + b];   // 8 keys of one statistics tile (128 / 256 keys)
            const float sc = __expf(tm - lse[b]);
            const int64_t lab = labels[b] - (int64_t)m_begin;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = m0 + m8 + e;
                float gv = bf2f(p[e]) * sc - ls * inv_m;
                gv -= ((int64_t)key == lab) ? 1.0f - ls : 0.f;
                g[e] = key < M ? gv : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) T[bl][m8 + e] = g[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = (t >> 3) + 32 * i, b8 = (t & 7) * 8;
        if (b0 + b8 >= B) continue;                                           // B % 8 == 0: whole vectors
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(T[b8 + e][ml]);
        *(bf16x8*)(Gt + (size_t)(m0 + ml) * B + b0 + b8) = o;
    }
}

int bank_stats_fwd(const BankArgs& a, float* stats, float* ws, size_t ws_bytes, hipStream_t st, float* zsave) {
    int rc = bank_check(a);
    if (rc) return rc;
    // per-GPU batches below 128 queries: the barrier-free row-tile kernel (bank2.hip); zsave (optional) keeps the logits
    // for bank_grad_q's saved path.  SPN_BANK2=0 keeps everything on the first-generation kernels below.
    if (bank_saved_path(a)) return bank2_stats_fwd(a, stats, zsave, ws, ws_bytes, st);
    // with a save buffer: ONE pass computes the statistics and the unnormalised query gradient (bank_fused_*)
    if (zsave && bank_fused_ok(a) && !(bank_saved_path_large(a) && !bank_fused_large())) return bank_fused_fwd(a, stats, zsave, st);
    {
        BankArgs b;
        size_t base;
        if (bank_expand_fp8(a, ws, ws_bytes, &b, &base, st)) return bank_stats_fwd(b, stats, ws, base, st, nullptr);
    }
    // Large batches: the logits pass as a 256x256-tile GEMM with a statistics epilogue (the bank is read once, the
    // queries come from L2); the streaming kernel below re-reads every bank tile once per 32 queries, which is the
    // right trade only while B is small (8-way data parallel: 32 per GPU).  SPN_BANK_GEMM=0 forces streaming.
    const bool use_gemm = bank_gemm_on();
    if (use_gemm && !a.bank_scale && !a.group && a.B >= 128 && a.D % 64 == 0) {
        const int nt = gemm_bank_stats_tiles(a.M, a.D);
        if (ws_bytes < (size_t)nt * a.B * 4 * sizeof(float)) return SPN_ERR_WORKSPACE;
        if (bank_s160_on() && bank_stats160_ok(a.B, a.D, a.ldq)) {
            // 128..256 queries: the 160-row-tile kernel (bank3.hip: every CU pulls, three bank k tiles in flight per CU); with a
            // save buffer it keeps p TRANSPOSED ([bank row][query], the layout of the dq GEMM's G^T operand)
            const int nt160 = bank_stats160_tiles(a.M);
            // this tiling writes more partial blocks than the 256-row GEMM tiling checked above: guard it by its own count
            if (ws_bytes < (size_t)nt160 * a.B * 4 * sizeof(float)) return SPN_ERR_WORKSPACE;
            {
                const double bytes = (double)a.M * a.D * 2 + (double)a.B * a.D * 2 + (double)a.B * 16;
                ProfScope prof(PK_BANK_FWD, bytes, st);
                if (zsave && bank_saved_path_large(a)) {
                    const LargeSave sv = large_save_at(zsave, a.B, a.M);
                    rc = bank_stats160(a.q, a.ldq, a.bank, a.labels, a.B, a.M, a.D, a.m_begin, a.inv_tau, ws, sv.P, sv.tmax, st);   // p^T [M][B] in the P region
                } else {
                    rc = bank_stats160(a.q, a.ldq, a.bank, a.labels, a.B, a.M, a.D, a.m_begin, a.inv_tau, ws, nullptr, nullptr, st);
                }
            }
            if (rc) return rc;
            hipLaunchKernelGGL(bank_stats_fold_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, ws, nt160, a.B, stats);
            SPN_CHECK_LAUNCH();
            return SPN_OK;
        }
        {
            const double bytes = (double)a.M * a.D * 2 + (double)a.B * a.D * 2 + (double)a.B * 16;
            ProfScope prof(PK_BANK_FWD, bytes, st);
            if (zsave && bank_saved_path_large(a)) {
                const LargeSave sv = large_save_at(zsave, a.B, a.M);
                rc = gemm_bank_stats(a.q, a.bank, a.B, a.M, a.D, a.ldq, a.D, a.labels, a.inv_tau, a.m_begin, ws, st, sv.P, sv.ldp,
                                     sv.tmax);
            } else {
                rc = gemm_bank_stats(a.q, a.bank, a.B, a.M, a.D, a.ldq, a.D, a.labels, a.inv_tau, a.m_begin, ws, st);
            }
        }
        if (rc) return rc;
        hipLaunchKernelGGL(bank_stats_fold_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, ws, nt, a.B, stats);
        SPN_CHECK_LAUNCH();
        return SPN_OK;
    }
    const BankChunking c = bank_chunking(a.B, a.M);
    if (ws_bytes < (size_t)c.nchunks * a.B * 4 * sizeof(float)) return SPN_ERR_WORKSPACE;
    if (tokmax_wave_path(a)) {
        rc = a.D == 128 ? launch_tokmax<128, false>(a, c, nullptr, 0.f, 0.f, ws, st)
                        : launch_tokmax<256, false>(a, c, nullptr, 0.f, 0.f, ws, st);
    } else if (a.bank_scale && !a.group && bank_fp8_mfma_on() && c.rows <= FP8_MAX_CHUNK_ROWS) {
        switch (a.D) {
            case 128: rc = launch_bank_fp8_fwd<128>(a, c, ws, st); break;
            case 256: rc = launch_bank_fp8_fwd<256>(a, c, ws, st); break;
            case 512: rc = launch_bank_fp8_fwd<512>(a, c, ws, st); break;
            case 640: rc = launch_bank_fp8_fwd<640>(a, c, ws, st); break;
            case 768: rc = launch_bank_fp8_fwd<768>(a, c, ws, st); break;
            case 1024: rc = launch_bank_fp8_fwd<1024>(a, c, ws, st); break;
            default: return SPN_ERR_SHAPE;
        }
    } else {
        SPN_BANK_DISPATCH(false, a, c, nullptr, 0.f, 0.f, ws, st)
    }
    if (rc) return rc;
    hipLaunchKernelGGL(bank_stats_fold_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, ws, c.nchunks, a.B, stats);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int bank_loss_finalize(const float* stats, int nshards, int B, int64_t M_total, float label_smoothing,
                       float* row_lse, float* row_loss, float* loss_mean, hipStream_t st) {
    if (nshards <= 0 || B <= 0 || M_total <= 0) return SPN_ERR_ARG;
    hipLaunchKernelGGL(bank_loss_finalize_kernel, dim3(1), dim3(256), 0, st, stats, nshards, B, 1.0f / (float)M_total,
                       label_smoothing, row_lse, row_loss, loss_mean);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int bank_grad_q(const BankArgs& a, const float* row_lse, float label_smoothing, int64_t M_total, float grad_scale,
                float* dq, float* ws, size_t ws_bytes, hipStream_t st, const float* zsaved) {
    int rc = bank_check(a);
    if (rc) return rc;
    if (!row_lse || !dq || M_total <= 0) return SPN_ERR_ARG;
    // logits saved by the forward call of this step: stream the bank once, no recomputation (bank2.hip)
    if (zsaved && bank_saved_path(a))
        return bank2_grad_q(a, zsaved, row_lse, label_smoothing, M_total, grad_scale, dq, ws, ws_bytes, st);
    if (zsaved && bank_fused_ok(a) && !(bank_saved_path_large(a) && !bank_fused_large())) {
        if (label_smoothing == 0.f) return bank_fused_bwd(a, zsaved, row_lse, grad_scale, dq, st);
        // label smoothing needs the column sums of the bank: this step's backward recomputes the logits (below)
    } else if (zsaved && bank_saved_path_large(a)) {
        const LargeSave sv = large_save_at(const_cast<float*>(zsaved), a.B, a.M);
        if (ws_bytes < gemm_tn_workspace_bytes(a.M, a.B, a.D)) return SPN_ERR_WORKSPACE;
        const double bytes = (double)a.M * a.D * 2 + (double)a.B * a.D * 6 + (double)a.B * 16;
        ProfScope prof(PK_BANK_BWD, bytes, st);
        if (bank_s160_on() && bank_stats160_ok(a.B, a.D, a.ldq)) {      // the forward call kept p^T [M][B]: scale it into G^T, no transpose
            rc = bank_gt_scale(sv.P, sv.Gt, sv.tmax, row_lse, a.labels, a.B, a.M, a.m_begin, label_smoothing, 1.0f / (float)M_total, st);
            if (rc) return rc;
            return gemm_tn(sv.Gt, a.bank, a.M, a.B, a.D, a.B, a.D, dq, a.D, grad_scale * a.inv_tau, 0, nullptr, ws, ws_bytes, st);
        }
        hipLaunchKernelGGL(bank_p_to_gt_kernel, dim3(sv.mpad / 64, (a.B + 63) / 64), dim3(256), 0, st, sv.P, sv.ldp, sv.tmax, row_lse,
                           a.labels, a.B, a.M, a.m_begin, label_smoothing, 1.0f / (float)M_total, sv.Gt, gemm_bank_stats_bn(a.D) == 128 ? 7 : 8);
        SPN_CHECK_LAUNCH();
        return gemm_tn(sv.Gt, a.bank, a.M, a.B, a.D, a.B, a.D, dq, a.D, grad_scale * a.inv_tau, 0, nullptr, ws, ws_bytes, st);
    }
    {
        BankArgs b;
        size_t base;
        if (bank_expand_fp8(a, ws, ws_bytes, &b, &base, st))
            return bank_grad_q(b, row_lse, label_smoothing, M_total, grad_scale, dq, ws, base, st, nullptr);
    }
    const BankChunking c = bank_chunking(a.B, a.M);
    if (ws_bytes < (size_t)c.nchunks * a.B * a.D * sizeof(float)) return SPN_ERR_WORKSPACE;
    if (tokmax_wave_path(a)) {
        rc = a.D == 128 ? launch_tokmax<128, true>(a, c, row_lse, label_smoothing, 1.0f / (float)M_total, ws, st)
                        : launch_tokmax<256, true>(a, c, row_lse, label_smoothing, 1.0f / (float)M_total, ws, st);
    } else {
        SPN_BANK_DISPATCH(true, a, c, row_lse, label_smoothing, 1.0f / (float)M_total, ws, st)
    }
    if (rc) return rc;
    return fold_rows(ws, (size_t)a.B * a.D, c.nchunks, (size_t)a.B * a.D, dq, grad_scale * a.inv_tau, 0, st);
}

// ------------------------------------------------------------------------------ fp8 bank
// One wave per row: scale = max|x| / 448 (the largest e4m3 value; 1 for an all-zero row), bytes = RNE(x / scale).
__global__ void bank_quantize_fp8_kernel(const float* __restrict__ bank, int M, int D, int Dp, uint8_t* __restrict__ out,
                                         float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* x = bank + (size_t)m * D;
    float mx = 0.f;
    for (int c = lane; c < D; c += 64) mx = fmaxf(mx, fabsf(x[c]));
    mx = wave_max(mx);
    const float sc = mx > 0.f ? mx / 448.0f : 1.0f;
    if (lane == 0) scale[m] = sc;
    for (int c = lane * 4; c < Dp; c += 256) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = c + e < D ? x[c + e] / sc : 0.f;
            v[e] = fminf(fmaxf(t, -448.0f), 448.0f);
        }
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        *(int*)(out + (size_t)m * Dp + c) = pk;
    }
}

int bank_quantize_fp8(const float* bank, int M, int D, int Dp, uint8_t* out, float* scale, hipStream_t st) {
    if (M <= 0 || D <= 0 || !bank || !out || !scale) return SPN_ERR_ARG;
    if (Dp % 4 || Dp < D) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(bank_quantize_fp8_kernel, dim3((M + 3) / 4), dim3(256), 0, st, bank, M, D, Dp, out, scale);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// -------------------------------------------------------------------- in-batch negatives
// The B x B case of the same loss with the TARGET side trainable too (clip4cir/models.py:160-167, labels =
// arange): the query gradient comes from bank_grad_q with the normalised targets as the bank; this is the
// other side, dt[j] = gs/tau * sum_i (exp(l_ij - lse_i) - [i == j]) q[i].  B is a batch size (<= a few
// thousand), so one block per target row with fp32 VALU dot products is far below every roofline that matters.
__global__ __launch_bounds__(256) void inbatch_grad_t_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ t,
                                                            int ldq, const float* __restrict__ row_lse, int B, int D,
                                                            float inv_tau, float gs, float* __restrict__ dt) {
    extern __shared__ float g[];   // [B] coefficients of this target row
    const int j = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bf16_t* tj = t + (size_t)j * ldq;
    for (int i = wid; i < B; i += 4) {
        const bf16_t* qi = q + (size_t)i * ldq;
        float s = 0.f;
        for (int d = lane * 4; d < D; d += 256) {
            const bf16x4 a = *(const bf16x4*)(qi + d), b = *(const bf16x4*)(tj + d);
            s += bf2f(a[0]) * bf2f(b[0]) + bf2f(a[1]) * bf2f(b[1]) + bf2f(a[2]) * bf2f(b[2]) + bf2f(a[3]) * bf2f(b[3]);
        }
        s = wave_sum(s);
        if (lane == 0) g[i] = (__expf(s * inv_tau - row_lse[i]) - (i == j ? 1.f : 0.f)) * gs * inv_tau;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int i = 0; i < B; ++i) acc += g[i] * bf2f(q[(size_t)i * ldq + d]);
        dt[(size_t)j * D + d] = acc;
    }
}

int inbatch_grad_t(const bf16_t* q, const bf16_t* t, int ldq, const float* row_lse, int B, int D, float inv_tau,
                   float grad_scale, float* dt, hipStream_t st) {
    if (B <= 0 || D <= 0) return SPN_ERR_ARG;
    if (D % 4 || ldq % 4 || ldq < D || (size_t)B * 4 > 64 * 1024) return SPN_ERR_SHAPE;
    hipLaunchKernelGGL(inbatch_grad_t_kernel, dim3(B), dim3(256), (size_t)B * 4, st, q, t, ldq, row_lse, B, D, inv_tau,
                       grad_scale, dt);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
