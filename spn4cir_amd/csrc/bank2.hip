// Bank InfoNCE, second generation: barrier-free streaming kernels for per-GPU batches below 128 queries
// (models_negplus.py:150-154 without the B x M logit matrix in fp32 autograd graphs; see bank.hip for the contract).
//
// Why a second pair of kernels.  bank_stream_kernel splits D over the four waves of a block, so every 32-row bank tile
// costs three block barriers (tile landed / partial logits exchanged / G ready) and its backward pass writes one
// [B, D] fp32 partial per block: 250 blocks x 98 KB = 24.6 MB at B = 32, M = 40 000 - 40 % on top of the 61 MB bank
// stream - which a separate fold launch reads back.  Measured 18 / 26 us (3.4 / 2.4 TB/s) for a pass whose only
// unavoidable traffic is the bank itself.  Here:
//
//   forward  (bank_rowtile_fwd_kernel): every WAVE owns whole bank rows (full D) and its own LDS ring, fed by its own
//     buffer_load ... lds pieces of 8 rows x 128 B; the 32 queries sit in registers (D = 768: 192 VGPRs, one wave per
//     SIMD).  A 16-row tile is 16 keys x 32 queries of v_mfma_f32_16x16x32_bf16 over the full D: no partial sums to
//     exchange, no block barrier in the loop - the only waits are the wave's own counted vmcnt, and the count is
//     uniform because the tail issues zero-fill pieces (out-of-range buffer offsets move no bytes).  The logits
//     z = <q, bank_j> / tau are SAVED (fp32 [B, M]: 5 MB at B = 32, 8 % of the bank bytes) for the backward pass.
//   backward (bank_dslice_bwd_kernel): with z saved nothing has to be recomputed, so a block no longer needs whole rows:
//     it owns ONE 128-column slice of D and a long row range (grid = D/128 slices x chunks = #CUs), its waves stream
//     32-row x 256 B tiles (+ the matching 32 x 32 block of z) through private rings and accumulate
//     dq[32, 128] += G^T bank in registers; partials shrink to chunks x B x D x 4 B = 4.1 MB (42 chunks) and the bank
//     is still read exactly once.
//
// The fp8 bank (BASELINE config 5) runs through the same kernels: the raw e4m3 rows go through the rings (half the
// bytes), fragments are converted to bf16 in registers behind the LDS read and the per-row scale multiplies the logit
// (forward) or G (backward) - the products are those of the bf16 MFMA path on the dequantised values.
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static constexpr int RQ = 32;          // queries per block

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 8 e4m3 bytes -> 8 bf16 (exact: every e4m3 value is a bf16 value)
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(uint32_t lo, uint32_t hi) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    return bf16x8{f2bf(a[0]), f2bf(a[1]), f2bf(b[0]), f2bf(b[1]), f2bf(c[0]), f2bf(c[1]), f2bf(d[0]), f2bf(d[1])};
}

__device__ __forceinline__ long pack_fp8x8_b2(const float (&v)[8]) {
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

__device__ __forceinline__ void unpack_fp8x8_b2(long p, float (&v)[8]) {
    const int lo = (int)(unsigned)((unsigned long)p & 0xffffffffu), hi = (int)(unsigned)((unsigned long)p >> 32);
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1]; v[6] = d[0]; v[7] = d[1];
}

// ---------------------------------------------------------------------------------------------- forward
// Ring piece = 8 rows x 128 B (1 KB, one DMA instruction); a ring BLOCK = two pieces = the 128-byte column block `i` of a
// 16-row tile (bf16: 64 k values = two MFMA k-steps; e4m3: 128 k values = four).  Inside a piece the 16-byte chunk c of
// row r sits at position c ^ ((r >> 1) & 7) (swizzle applied on the SOURCE address of the DMA, which writes LDS
// linearly): the 16 rows of a fragment read then cover all 16-byte bank groups (rows alternate between the two halves of
// a 256-byte bank line, the XOR spreads the eight even / odd rows over its eight 16-byte slots).
struct RowTileGeom {
    int nq, nchunks, rows_per_block;
};

static constexpr int RT_LEAD = 17;                 // ring blocks in flight per wave (34 KB)
static constexpr int RT_RING = RT_LEAD + 1;        // + the block being consumed
static constexpr int RT_MAX_ROWS_PER_WAVE = 512;   // fp8: a wave's row scales live in LDS (2 KB)

static RowTileGeom rowtile_geom(int B, int M) {
    RowTileGeom g;
    g.nq = (B + RQ - 1) / RQ;
    int target = device_cu_count() / g.nq;
    if (target < 1) target = 1;
    int rows = (M + target - 1) / target;
    rows = (rows + 31) / 32 * 32;                  // 4 waves x a multiple of 8 rows
    if (rows > 4 * RT_MAX_ROWS_PER_WAVE) rows = 4 * RT_MAX_ROWS_PER_WAVE;
    g.rows_per_block = rows;
    g.nchunks = (M + rows - 1) / rows;
    return g;
}

template <int D, bool FP8, bool SAVE>
__global__ __launch_bounds__(256, 1) void bank_rowtile_fwd_kernel(BankArgs a, RowTileGeom gm, float* __restrict__ zsave, int ldz,
                                                                 float* __restrict__ ws) {
    constexpr int EB = FP8 ? 1 : 2;                   // bytes per bank element
    constexpr int ROWB = D * EB;                      // bytes per bank row
    constexpr int NBT = ROWB / 128;                   // ring blocks per 16-row tile
    constexpr int KPB = FP8 ? 4 : 2;                  // MFMA k-steps (32 k values) per ring block
    constexpr int KS = D / 32;
    static_assert(ROWB % 128 == 0 && KS == NBT * KPB, "bank width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + w * (RT_RING * 2048);
    float* Fin = (float*)(smem + 4 * RT_RING * 2048);                 // [4 waves][RQ][4]
    float* Ssc = Fin + 4 * RQ * 4 + w * RT_MAX_ROWS_PER_WAVE;         // fp8: this wave's row scales
    const int qi = blockIdx.x / gm.nchunks, mi = blockIdx.x % gm.nchunks;
    const int q0 = qi * RQ;
    const int rw = gm.rows_per_block >> 2;                            // rows per wave, a multiple of 8
    const int r_lo = mi * gm.rows_per_block + w * rw;                 // shard-local first row of this wave
    const int r_hi = min(a.M, r_lo + rw);
    const int ntiles = r_hi > r_lo ? (r_hi - r_lo + 15) >> 4 : 0;
    const int nblk = ntiles * NBT;
    // rows >= r_hi read as zero and move no bytes (the half-tile at the end of a wave's range, the uniform tail pieces)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bank, (uint32_t)max(r_hi, 0) * (uint32_t)ROWB);

    // the 32 queries of the block, full D, in registers (B operand: j = query, k = d).  e4m3 bank: the logits run on
    // v_mfma_f32_16x16x32_fp8_fp8 straight from the raw tile, which needs fp8 queries too - each query is split once
    // into two e4m3 terms with one scale, q ~= sq * hi + (sq / 16) * lo (bank.hip: bank_fp8_fwd_kernel;
    // oracle/bank_loss.py split_query_e4m3 restates it bit for bit): logit = sb[key] * sq * (acc_hi + acc_lo / 16) / tau.
    [[maybe_unused]] bf16x8 qf[FP8 ? 1 : 2][FP8 ? 1 : KS];
    [[maybe_unused]] long qh[FP8 ? 2 : 1][FP8 ? KS : 1], ql[FP8 ? 2 : 1][FP8 ? KS : 1];
    [[maybe_unused]] float sq[2] = {1.f, 1.f};
    int64_t label[2];
    bool q_ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int qr = q0 + mt * 16 + (lane & 15);
        q_ok[mt] = qr < a.B;
        label[mt] = q_ok[mt] ? a.labels[qr] - (int64_t)a.m_begin : -1;
        if constexpr (!FP8) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (q_ok[mt]) {
                    qf[mt][ks] = *(const bf16x8*)(a.q + (size_t)qr * a.ldq + ks * 32 + (lane >> 4) * 8);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[mt][ks][e] = (bf16_t)0.0f;
                }
            }
        } else {
            bf16x8 raw[KS];
            float am = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (q_ok[mt]) {
                    raw[ks] = *(const bf16x8*)(a.q + (size_t)qr * a.ldq + ks * 32 + (lane >> 4) * 8);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) raw[ks][e] = (bf16_t)0.0f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(bf2f(raw[ks][e])));
            }
            am = fmaxf(am, __shfl_xor(am, 16, 64));          // the four lanes l, l^16, l^32, l^48 share a query row
            am = fmaxf(am, __shfl_xor(am, 32, 64));
            sq[mt] = am > 0.f ? am / 448.0f : 1.0f;
            const float rh = 1.0f / sq[mt], rl = rh * 16.0f; // separate multiply / subtract (no fma): as the oracle's model
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float v[8], hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(__fmul_rn(bf2f(raw[ks][e]), rh), -448.0f), 448.0f);
                qh[mt][ks] = pack_fp8x8_b2(v);
                unpack_fp8x8_b2(qh[mt][ks], hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float res = __fsub_rn(bf2f(raw[ks][e]), __fmul_rn(hv[e], sq[mt]));
                    v[e] = fminf(fmaxf(__fmul_rn(res, rl), -448.0f), 448.0f);
                }
                ql[mt][ks] = pack_fp8x8_b2(v);
            }
        }
    }
    if constexpr (FP8) {
        for (int i = lane; i < r_hi - r_lo; i += 64) Ssc[i] = a.bank_scale[r_lo + i];
    }
    wait_vm0();                                       // queries (and scales) are in: the counted waits below see DMA only

    auto issue = [&](int jb) {                        // ring block jb of this wave's sequence (wave-uniform)
        const int t = jb / NBT, i = jb - t * NBT;
        char* dst = ring + (jb % RT_RING) * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = t * 16 + h * 8 + (lane >> 3);                      // row inside this wave's range
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            uint32_t off = (uint32_t)(r_lo + row) * (uint32_t)ROWB + (uint32_t)(i * 128 + c * 16);
            if (jb >= nblk) off = 0xFFFFFFF0u;                                 // tail: zero fill, no memory traffic
            glds16(rs, dst + h * 1024, off);
        }
    };

    float st_m[2] = {-INFINITY, -INFINITY}, st_l[2] = {0.f, 0.f}, st_sl[2] = {0.f, 0.f}, st_lab[2] = {-INFINITY, -INFINITY};
    for (int jb = 0; jb < RT_LEAD; ++jb) issue(jb);
    const int arow = lane & 15, acol = lane >> 4;
    const int aswz = (arow >> 1) & 7;
    const uint32_t abase = (uint32_t)((arow >> 3) * 1024 + (arow & 7) * 128);
    for (int t = 0; t < ntiles; ++t) {
        f32x4 s[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        [[maybe_unused]] f32x4 s2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};      // e4m3: the low query term
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            const int jb = t * NBT + i;
            // the slot refilled now was read one block ago: its fragment reads have returned (their MFMAs were issued)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            issue(jb + RT_LEAD);
            wait_vmcnt<2 * RT_LEAD>();                // block jb has landed; RT_LEAD newer blocks stay in flight
            __builtin_amdgcn_sched_barrier(0);
            const char* blk = ring + (jb % RT_RING) * 2048 + abase;
#pragma unroll
            for (int kk = 0; kk < KPB; ++kk) {
                if constexpr (FP8) {
                    const int u = kk * 4 + acol;                               // 8-byte unit of the 128-byte row
                    const long af = *(const long*)(blk + (((u >> 1) ^ aswz) << 4) + (u & 1) * 8);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        s[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af, qh[mt][i * KPB + kk], s[mt], 0, 0, 0);
                        s2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(af, ql[mt][i * KPB + kk], s2[mt], 0, 0, 0);
                    }
                } else {
                    const bf16x8 af = *(const bf16x8*)(blk + (((kk * 4 + acol) ^ aswz) << 4));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) s[mt] = mfma16(af, qf[mt][i * KPB + kk], s[mt]);
                }
            }
        }
        // lane: query (lane & 15) of each mt, keys key0 .. key0 + 3 of this tile
        const int key0 = r_lo + t * 16 + acol * 4;
        const bool live4 = key0 < r_hi;               // r_hi - r_lo is a multiple of 8 except at the end of the shard
        f32x4 sc4 = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FP8) sc4 = *(const f32x4*)(Ssc + t * 16 + acol * 4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 v;
            float tm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = key0 + r < r_hi;
                float acc = s[mt][r];
                if constexpr (FP8) acc = (acc + s2[mt][r] * 0.0625f) * sq[mt];
                v[r] = live ? acc * sc4[r] * a.inv_tau : -INFINITY;
                tm = fmaxf(tm, v[r]);
            }
            if (tm > -INFINITY) {
                const float mn = fmaxf(st_m[mt], tm);
                float add = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = key0 + r < r_hi;
                    add += live ? __expf(v[r] - mn) : 0.f;
                    st_sl[mt] += live ? v[r] : 0.f;
                    st_lab[mt] = (live && (int64_t)(key0 + r) == label[mt]) ? v[r] : st_lab[mt];
                }
                st_l[mt] = st_l[mt] * __expf(st_m[mt] - mn) + add;
                st_m[mt] = mn;
            }
            if constexpr (SAVE) {
                // ldz is a multiple of 32: the vector store stays inside the row; keys beyond the shard hold -inf
                if (live4 && q_ok[mt]) *(f32x4*)(zsave + (size_t)(q0 + mt * 16 + arow) * ldz + key0) = v;
            }
        }
    }
    // merge the four key groups of a query (lanes l, l^16, l^32, l^48), then the four waves through LDS
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(st_m[mt], off, 64), l2 = __shfl_xor(st_l[mt], off, 64);
            const float mn = fmaxf(st_m[mt], m2);
            st_l[mt] = mn > -INFINITY ? st_l[mt] * __expf(st_m[mt] - mn) + l2 * __expf(m2 - mn) : 0.f;
            st_m[mt] = mn;
            st_sl[mt] += __shfl_xor(st_sl[mt], off, 64);
            st_lab[mt] = fmaxf(st_lab[mt], __shfl_xor(st_lab[mt], off, 64));
        }
        if (lane < 16) *(f32x4*)(Fin + (w * RQ + mt * 16 + lane) * 4) = f32x4{st_m[mt], st_l[mt], st_sl[mt], st_lab[mt]};
    }
    wait_vm0();                                       // the zero-fill tail pieces still target this wave's ring
    __syncthreads();
    if (tid < RQ) {
        float m = -INFINITY, l = 0.f, sl = 0.f, lab = -INFINITY;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const f32x4 p = *(const f32x4*)(Fin + (ww * RQ + tid) * 4);
            const float mn = fmaxf(m, p[0]);
            if (mn > -INFINITY) l = l * __expf(m - mn) + p[1] * __expf(p[0] - mn);
            m = mn;
            sl += p[2];
            lab = fmaxf(lab, p[3]);
        }
        const int q = q0 + tid;
        if (q < a.B) *(f32x4*)(ws + ((size_t)mi * a.B + q) * 4) = f32x4{m, l, sl, lab};
    }
}

template <int D, bool FP8, bool SAVE>
static int launch_rowtile_fwd(const BankArgs& a, const RowTileGeom& g, float* zsave, int ldz, float* ws, hipStream_t st) {
    const size_t lds = 4 * (size_t)RT_RING * 2048 + 4 * RQ * 4 * sizeof(float) + (FP8 ? 4 * RT_MAX_ROWS_PER_WAVE * sizeof(float) : 0);
    auto kern = bank_rowtile_fwd_kernel<D, FP8, SAVE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        const double bytes = (double)a.M * D * (FP8 ? 1 : 2) + (FP8 ? 4.0 * a.M : 0.0) + (double)a.B * D * 2 + (double)a.B * 16;
        ProfScope prof(PK_BANK_FWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(g.nq * g.nchunks), dim3(256), lds, st, a, g, zsave, ldz, ws);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// --------------------------------------------------------------------------------------------- backward
// Block = (query tile, row chunk, 128-column slice of D); its four waves take the chunk's 32-row tiles round robin,
// each through a private ring of 3 slots x (32 rows x 256 B of the bank slice + the 32 queries x 32 keys block of the
// saved logits, fp32).  Bank slot image: row r at byte r * 256 (fp8: r * 128 raw bytes, converted below), 16-byte chunk c
// at position c ^ bank_swz(r & 15) - the layout bank_stream_kernel uses for its transpose reads (same row alignment);
// z slot image: query row q at byte q * 128, chunk c (4 keys) at position c ^ ((q >> 1) & 7).
struct DSliceGeom {
    int nq, nch, rows_per_chunk, nsl;
};

static constexpr int DS_SLOTS = 3;
static constexpr int DS_Z_B = 32 * 128;
// slot = bank part (bf16: 32 rows x 256 B; e4m3: 32 x 128 B raw) + z block (4 KB) + (e4m3) the tile's 32 row scales (1 KB piece)
template <bool FP8> struct DSlot {
    static constexpr int BANK_B = FP8 ? 32 * 128 : 32 * 256;
    static constexpr int SLOT_B = BANK_B + DS_Z_B + (FP8 ? 1024 : 0);
};
static constexpr int DS_CVT_B = 32 * 256;                  // e4m3: a wave's bf16 image of its current tile

__device__ __forceinline__ int bank2_swz(int r) {          // == bank_swz (bank.hip)
    return (((r & 3) | (((r >> 3) & 1) << 2)) << 1) | ((r >> 2) & 1);
}

static DSliceGeom dslice_geom(int B, int M, int D) {
    DSliceGeom g;
    g.nq = (B + RQ - 1) / RQ;
    g.nsl = D / 128;
    int target = device_cu_count() / (g.nq * g.nsl);
    if (target < 1) target = 1;
    int rows = (M + target - 1) / target;
    rows = (rows + 127) / 128 * 128;                       // 4 waves x 32-row tiles
    g.rows_per_chunk = rows;
    g.nch = (M + rows - 1) / rows;
    return g;
}

template <int D, bool FP8>
__global__ __launch_bounds__(256, 1) void bank_dslice_bwd_kernel(BankArgs a, DSliceGeom gm, const float* __restrict__ zs, int ldz,
                                                                const float* __restrict__ row_lse, float label_smoothing,
                                                                float inv_m_total, float* __restrict__ ws) {
    constexpr int EB = FP8 ? 1 : 2;
    constexpr int ROWB = D * EB;
    constexpr int NBK = FP8 ? 4 : 8;                       // DMA instructions of a tile's bank part
    constexpr int NDMA = NBK + 4 + (FP8 ? 1 : 0);          // + the z block (+ the row scales)
    constexpr int DS_BANK_B = DSlot<FP8>::BANK_B, DS_SLOT_B = DSlot<FP8>::SLOT_B;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + w * (DS_SLOTS * DS_SLOT_B);
    char* cvt = smem + 4 * DS_SLOTS * DS_SLOT_B + w * DS_CVT_B;        // fp8: this wave's bf16 image of the current tile
    const int per_q = gm.nch * gm.nsl;
    const int qi = blockIdx.x / per_q, rem = blockIdx.x % per_q;
    const int ci = rem / gm.nsl, sl = rem % gm.nsl;                    // the slices of a chunk are neighbours in the grid
    const int q0 = qi * RQ;
    const int m_lo = ci * gm.rows_per_chunk;
    const int m_hi = min(a.M, m_lo + gm.rows_per_chunk);
    const int ntiles = m_hi > m_lo ? (m_hi - m_lo + 31) >> 5 : 0;
    const int nmine = ntiles > w ? (ntiles - w + 3) >> 2 : 0;
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(a.bank, (uint32_t)a.M * (uint32_t)ROWB);
    const __amdgpu_buffer_rsrc_t rsz = make_rsrc(zs, (uint32_t)a.B * (uint32_t)ldz * 4u);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rss = make_rsrc(FP8 ? (const void*)a.bank_scale : (const void*)zs,
                                                                  FP8 ? (uint32_t)a.M * 4u : 0u);

    float lse[2];
    int64_t label[2];
    bool q_ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int qr = q0 + mt * 16 + (lane & 15);
        q_ok[mt] = qr < a.B;
        label[mt] = q_ok[mt] ? a.labels[qr] - (int64_t)a.m_begin : -1;
        lse[mt] = q_ok[mt] ? row_lse[qr] : 0.f;
    }
    wait_vm0();
    f32x4 dq[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) dq[mt][dt] = f32x4{0, 0, 0, 0};

    auto issue = [&](int i) {                              // this wave's i-th tile (wave-uniform); i >= nmine: zero fill
        const int mrow0 = m_lo + (w + 4 * i) * 32;
        char* dst = ring + (i % DS_SLOTS) * DS_SLOT_B;
        const bool dead = i >= nmine;
#pragma unroll
        for (int ii = 0; ii < NBK; ++ii) {
            uint32_t off;
            if constexpr (FP8) {                           // 8 rows x 128 raw bytes per instruction, linear image
                const int r = ii * 8 + (lane >> 3);
                off = (uint32_t)(mrow0 + r) * (uint32_t)ROWB + (uint32_t)(sl * 128 + (lane & 7) * 16);
            } else {                                       // 4 rows x 256 B per instruction, swizzled image
                const int r = ii * 4 + (lane >> 4);
                const int c = (lane & 15) ^ bank2_swz(r & 15);
                off = (uint32_t)(mrow0 + r) * (uint32_t)ROWB + (uint32_t)(sl * 256 + c * 16);
            }
            if (dead) off = 0xFFFFFFF0u;
            glds16(rsb, dst + ii * 1024, off);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {                   // z: 8 query rows x 32 keys (128 B) per instruction
            const int q = ii * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((q >> 1) & 7);
            uint32_t off = ((uint32_t)(q0 + q) * (uint32_t)ldz + (uint32_t)(mrow0 + c * 4)) * 4u;
            if (dead || q0 + q >= a.B) off = 0xFFFFFFF0u;
            glds16(rsz, dst + DS_BANK_B + ii * 1024, off);
        }
        if constexpr (FP8) {                               // the tile's 32 row scales (lanes 0..7; rows >= M read 0)
            uint32_t off = (uint32_t)(mrow0 + lane * 4) * 4u;
            if (dead || lane >= 8) off = 0xFFFFFFF0u;
            glds16(rss, dst + DS_BANK_B + DS_Z_B, off);
        }
    };

    issue(0);
    issue(1);
    for (int i = 0; i < nmine; ++i) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot refilled now was read one tile ago
        __builtin_amdgcn_sched_barrier(0);
        issue(i + 2);
        wait_vmcnt<2 * NDMA>();
        __builtin_amdgcn_sched_barrier(0);
        const char* T = ring + (i % DS_SLOTS) * DS_SLOT_B;
        const char* Z = T + DS_BANK_B;
        const int mrow0 = m_lo + (w + 4 * i) * 32;
        // G operand (B: j = query, k = key (lane >> 4) * 8 + e) from the saved logits
        bf16x8 gf[2];
        f32x4 scl[2] = {f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{1.f, 1.f, 1.f, 1.f}};
        if constexpr (FP8) {                               // dq = sum_key (G * scale_key) * e4m3(key, :)
            const float* sp = (const float*)(Z + DS_Z_B) + (lane >> 4) * 8;
            scl[0] = *(const f32x4*)sp;
            scl[1] = *(const f32x4*)(sp + 4);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int q = mt * 16 + (lane & 15);
            const int zsw = (q >> 1) & 7;
            const f32x4 z0 = *(const f32x4*)(Z + q * 128 + ((((lane >> 4) * 2) ^ zsw) << 4));
            const f32x4 z1 = *(const f32x4*)(Z + q * 128 + ((((lane >> 4) * 2 + 1) ^ zsw) << 4));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = mrow0 + (lane >> 4) * 8 + e;
                const float z = e < 4 ? z0[e] : z1[e - 4];
                float gv = __expf(z - lse[mt]) - label_smoothing * inv_m_total;
                gv -= ((int64_t)key == label[mt]) ? 1.0f - label_smoothing : 0.f;
                gv = (q_ok[mt] && key < m_hi) ? gv : 0.f;
                if constexpr (FP8) gv *= scl[e >> 2][e & 3];
                gf[mt][e] = f2bf(gv);
            }
        }
        const char* Tb = T;
        if constexpr (FP8) {
            // raw e4m3 [32][128 B] -> this wave's swizzled bf16 image [32][256 B] (exact conversion); wave-private, so the
            // LDS round trip needs no barrier, only the wave's own lgkmcnt
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int r = ii * 8 + (lane >> 3), cb = lane & 7;             // 16 raw bytes = columns 16 cb .. + 15
                const u32x4 v = *(const u32x4*)(T + r * 128 + cb * 16);
                const int sw = bank2_swz(r & 15);
                *(bf16x8*)(cvt + r * 256 + (((2 * cb) ^ sw) << 4)) = fp8x8_to_bf16(v[0], v[1]);
                *(bf16x8*)(cvt + r * 256 + (((2 * cb + 1) ^ sw) << 4)) = fp8x8_to_bf16(v[2], v[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            Tb = cvt;
        }
        // dq[q][d] += sum_key G[q][key] bank[key][d]:  D[i = d][j = query], k = key (one 32-step per tile)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
                const int col = dt * 16 + (lane & 3) * 4;
                u.h[h] = lds_tr16_b64(Tb + r * 256 + (((col >> 3) ^ bank2_swz(r & 15)) << 4) + (col & 7) * 2);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) dq[mt][dt] = mfma16(u.v, gf[mt], dq[mt][dt]);
        }
    }
    // the four waves' partials -> LDS (the rings are idle now) -> one [32, 128] slab of this chunk's partial
    wait_vm0();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    float* mine = (float*)(smem + w * (RQ * 128 * 4));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
            *(f32x4*)(mine + (mt * 16 + (lane & 15)) * 128 + dt * 16 + (lane >> 4) * 4) = dq[mt][dt];
    __syncthreads();
    for (int e = tid * 4; e < RQ * 128; e += 256 * 4) {
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) acc += *(const f32x4*)((const float*)(smem + ww * (RQ * 128 * 4)) + e);
        const int q = q0 + e / 128;
        if (q < a.B) *(f32x4*)(ws + ((size_t)ci * a.B + q) * D + sl * 128 + (e & 127)) = acc;
    }
}

template <int D, bool FP8>
static int launch_dslice_bwd(const BankArgs& a, const DSliceGeom& g, const float* zs, int ldz, const float* row_lse, float ls,
                             float inv_m, float* ws, hipStream_t st) {
    const size_t lds = 4 * (size_t)DS_SLOTS * DSlot<FP8>::SLOT_B + (FP8 ? 4 * (size_t)DS_CVT_B : 0);
    auto kern = bank_dslice_bwd_kernel<D, FP8>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        const double bytes = (double)a.M * D * (FP8 ? 1 : 2) + (FP8 ? 4.0 * a.M : 0.0) + (double)a.B * D * 6 + (double)a.B * 16;
        ProfScope prof(PK_BANK_BWD, bytes, st);
        hipLaunchKernelGGL(kern, dim3(g.nq * g.nch * g.nsl), dim3(256), lds, st, a, g, zs, ldz, row_lse, ls, inv_m, ws);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------------------------------ host
static bool bank2_on() {
    static const bool on = [] {
        const char* e = getenv("SPN_BANK2");
        return !(e && e[0] == '0');
    }();
    return on;
}

// The saved-logits pair serves plain (one row = one target) banks, bf16 or e4m3, at per-call batches below 128 queries
bool bank_saved_path(const BankArgs& a) {
    if (!bank2_on() || a.group || a.B >= 128 || a.B <= 0) return false;
    switch (a.D) {
        case 128: case 256: case 512: case 640: case 768: case 1024: return true;
        default: return false;
    }
}

int bank_saved_ld(int M) { return (M + 31) / 32 * 32; }

size_t bank_saved_bytes(int B, int M) { return (size_t)B * bank_saved_ld(M) * sizeof(float); }

size_t bank2_workspace_bytes(int B, int M, int D) {
    if (B >= 128 || D % 128) return 0;
    const RowTileGeom f = rowtile_geom(B, M);
    const DSliceGeom b = dslice_geom(B, M, D);
    const size_t s = (size_t)f.nchunks * B * 4 * sizeof(float), d = (size_t)b.nch * B * D * sizeof(float);
    return s > d ? s : d;
}

#define SPN_B2_CASE(D_, CALL_)                          \
    case D_: rc = a.bank_scale ? CALL_(D_, true) : CALL_(D_, false); break;

int bank2_stats_fwd(const BankArgs& a, float* stats, float* zsave, float* ws, size_t ws_bytes, hipStream_t st) {
    const RowTileGeom g = rowtile_geom(a.B, a.M);
    if (ws_bytes < (size_t)g.nchunks * a.B * 4 * sizeof(float)) return SPN_ERR_WORKSPACE;
    const int ldz = bank_saved_ld(a.M);
    int rc = SPN_ERR_SHAPE;
#define SPN_B2_FWD(D_, F8_) (zsave ? launch_rowtile_fwd<D_, F8_, true>(a, g, zsave, ldz, ws, st) \
                                   : launch_rowtile_fwd<D_, F8_, false>(a, g, nullptr, ldz, ws, st))
    switch (a.D) {
        SPN_B2_CASE(128, SPN_B2_FWD)
        SPN_B2_CASE(256, SPN_B2_FWD)
        SPN_B2_CASE(512, SPN_B2_FWD)
        SPN_B2_CASE(640, SPN_B2_FWD)
        SPN_B2_CASE(768, SPN_B2_FWD)
        SPN_B2_CASE(1024, SPN_B2_FWD)
        default: return SPN_ERR_SHAPE;
    }
#undef SPN_B2_FWD
    if (rc) return rc;
    return bank_stats_fold(ws, g.nchunks, a.B, stats, st);
}

int bank2_grad_q(const BankArgs& a, const float* zsaved, const float* row_lse, float label_smoothing, int64_t M_total,
                 float grad_scale, float* dq, float* ws, size_t ws_bytes, hipStream_t st) {
    const DSliceGeom g = dslice_geom(a.B, a.M, a.D);
    if (ws_bytes < (size_t)g.nch * a.B * a.D * sizeof(float)) return SPN_ERR_WORKSPACE;
    const int ldz = bank_saved_ld(a.M);
    const float inv_m = 1.0f / (float)M_total;
    int rc = SPN_ERR_SHAPE;
#define SPN_B2_BWD(D_, F8_) launch_dslice_bwd<D_, F8_>(a, g, zsaved, ldz, row_lse, label_smoothing, inv_m, ws, st)
    switch (a.D) {
        SPN_B2_CASE(128, SPN_B2_BWD)
        SPN_B2_CASE(256, SPN_B2_BWD)
        SPN_B2_CASE(512, SPN_B2_BWD)
        SPN_B2_CASE(640, SPN_B2_BWD)
        SPN_B2_CASE(768, SPN_B2_BWD)
        SPN_B2_CASE(1024, SPN_B2_BWD)
        default: return SPN_ERR_SHAPE;
    }
#undef SPN_B2_BWD
    if (rc) return rc;
    return fold_rows(ws, (size_t)a.B * a.D, g.nch, (size_t)a.B * a.D, dq, grad_scale * a.inv_tau, 0, st);
}

}  // namespace spn
