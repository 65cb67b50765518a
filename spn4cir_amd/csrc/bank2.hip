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
//     buffer_load ... lds instructions of 1 KB of consecutive bytes; the 32 queries sit in registers (D = 768: 192
//     VGPRs, one wave per SIMD).  A tile is 8 keys x 32 queries of v_mfma_f32_16x16x32_bf16 over the full D: no partial
//     sums to exchange, no block barrier in the loop - the only waits are the wave's own counted vmcnt, and the count is
//     uniform because the tail issues zero-fill tiles (out-of-range buffer offsets move no bytes).  The logits
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

// The row-tile forward keeps its 32 queries (192 registers at D = 768) in ACCUMULATION registers: as ordinary values
// they fill the 256 architectural VGPRs, and the compiler then serialises every fragment read behind the previous
// step's MFMAs through one 4-register temporary (read -> wait -> MFMA, ~150 cycles per k-step).  MFMA operands may
// live in AGPRs on gfx950 (unified 512-entry file), so the accumulator and the B operand are pinned there by constraint
// and the queries are parked in AGPRs once at start-up (at D = 1024 they alone are 256 registers: the fragments beyond the
// budget stay in VGPRs and use the "v" form - left to the allocator, the overflow was copied into a temporary AGPR in
// front of every use, a VALU write -> XDL read hazard nobody pads).  The hazard recogniser does not look into inline asm: the caller puts >= 18 wait states (s_nop) between
// the last MFMA of a chain and the first VALU read of its accumulator (8-pass XDL write -> VALU read).
__device__ __forceinline__ void mfma16_aq(f32x4& acc, const bf16x8& a, const bf16x8& q) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(q));
}
__device__ __forceinline__ void mfma16_fp8_aq(f32x4& acc, long a, long q) {
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(q));
}
// the same with the B operand in a VGPR: the few query fragments that do not fit the 256 AGPRs (D = 1024)
__device__ __forceinline__ void mfma16_vq(f32x4& acc, const bf16x8& a, const bf16x8& q) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(q));
}
__device__ __forceinline__ void mfma16_fp8_vq(f32x4& acc, long a, long q) {
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(q));
}
// AGPR budget of the parked queries: 256 minus the accumulators (and a little air); register index of value (mt, ks[, term])
__host__ __device__ constexpr bool q_parked(int first_reg, int nregs) { return first_reg + nregs <= 236; }
// the accumulators are operands of the drain, so no read of them can be scheduled in front of it
__device__ __forceinline__ void mfma_drain(f32x4& a, f32x4& b) { asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a), "+a"(b)); }
__device__ __forceinline__ void mfma_drain(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a), "+a"(b), "+a"(c), "+a"(d));
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
// A wave's rows are one contiguous byte range of the bank; it streams them in TILES of 8 rows (bf16, D = 768: 12 KB),
// every DMA instruction moving 1 KB of CONSECUTIVE bytes (scattered 128-byte pieces of 8 different rows per instruction
// measured 4.4 TB/s at any bank size, consecutive kilobytes are what the HBM channels want).  LDS image of a tile = its
// rows back to back (row r at r * ROWB); inside each aligned group of 16 (bf16 rows are multiples of 256 B: every row
// starts on bank 0) resp. 8 sixteen-byte chunks, chunk c sits at position c ^ x(r), x = 2 (r & 7) resp. r & 7 - applied
// to the SOURCE address of the DMA (still inside the same 1 KB run), so that the eight rows of a fragment read, and the
// two neighbouring chunks the lane quarters of a wave read together, fall on distinct bank groups.
// An 8-row tile feeds the 16-row A operand of v_mfma_f32_16x16x32 with its rows twice (lanes 8-15 re-read rows 0-7, an
// LDS broadcast); the duplicate output rows are ignored.  That doubles the MFMA count per byte - 48 per 12 KB and wave,
// a fifth of what the HBM stream leaves room for - and halves the LDS a ring slot needs: three slots per wave, two tiles
// in flight behind the one being consumed.
struct RowTileGeom {
    int nq, nchunks, rows_per_block;
    int dbg;      // SPN_BANK2_DBG (experiments; results wrong): 1 = stream only (no fragment reads / MFMA / statistics)
};

static constexpr int RT_RING_B = 36 * 1024;        // LDS ring of one wave
static constexpr int RT_MAX_ROWS_PER_WAVE = 512;   // fp8: a wave's row scales live in LDS (2 KB)

static RowTileGeom rowtile_geom(int B, int M) {
    RowTileGeom g;
    g.nq = (B + RQ - 1) / RQ;
    int target = device_cu_count() / g.nq;
    if (target < 1) target = 1;
    int rows = (M + target - 1) / target;
    rows = (rows + 31) / 32 * 32;                  // 4 waves x whole 8-row tiles
    if (rows > 4 * RT_MAX_ROWS_PER_WAVE) rows = 4 * RT_MAX_ROWS_PER_WAVE;
    g.rows_per_block = rows;
    g.nchunks = (M + rows - 1) / rows;
    static const int dbg = [] { const char* e = spn_env("SPN_BANK2_DBG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
    return g;
}

template <int D, bool FP8, bool SAVE>
__global__ __launch_bounds__(256, 1) void bank_rowtile_fwd_kernel(BankArgs a, RowTileGeom gm, float* __restrict__ zsave, int ldz,
                                                                 float* __restrict__ ws) {
    constexpr int EB = FP8 ? 1 : 2;                   // bytes per bank element
    constexpr int ROWB = D * EB;                      // bytes per bank row
    constexpr int TILE_B = 8 * ROWB;                  // one tile: 8 consecutive rows
    constexpr int NP = TILE_B / 1024;                 // DMA instructions per tile
    constexpr int NS0 = RT_RING_B / TILE_B;
    constexpr int NS = (NS0 - 1) * NP > 60 ? 60 / NP + 1 : NS0;      // ring slots (vmcnt counts at most 63)
    constexpr int KS = D / 32;
    constexpr int XG = ROWB % 256 == 0 ? 16 : 8;      // chunks per swizzle group
    static_assert(TILE_B % 1024 == 0 && NS >= 2 && ROWB % 128 == 0, "bank width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + w * RT_RING_B;
    float* Fin = (float*)(smem + 4 * RT_RING_B);                      // [4 waves][RQ][4]
    float* Ssc = Fin + 4 * RQ * 4 + w * RT_MAX_ROWS_PER_WAVE;         // fp8: this wave's row scales
    const int qi = blockIdx.x / gm.nchunks, mi = blockIdx.x % gm.nchunks;
    const int q0 = qi * RQ;
    const int rw = gm.rows_per_block >> 2;                            // rows per wave, a multiple of 8
    const int r_lo = mi * gm.rows_per_block + w * rw;                 // shard-local first row of this wave
    const int r_hi = min(a.M, r_lo + rw);
    const int ntiles = r_hi > r_lo ? (r_hi - r_lo + 7) >> 3 : 0;
    // rows >= r_hi read as zero and move no bytes (the last tile of the shard, the uniform tail tiles)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bank, (uint32_t)max(r_hi, 0) * (uint32_t)ROWB);

    auto issue = [&](int t) {                         // tile t of this wave (wave-uniform); t >= ntiles: zero fill
        char* dst = ring + (t % NS) * TILE_B;
        const uint32_t base = (uint32_t)(r_lo + t * 8) * (uint32_t)ROWB;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int pos = p * 1024 + lane * 16;                              // LDS byte inside the tile image
            const int r = pos / ROWB, cp = (pos - r * ROWB) >> 4;              // row, chunk position
            const int c = cp ^ (XG == 16 ? 2 * r : r);                         // logical chunk fetched into it (r < 8)
            uint32_t off = base + (uint32_t)(r * ROWB + c * 16);
            if (t >= ntiles) off = 0xFFFFFFF0u;
            glds16(rs, dst + p * 1024, off);
        }
    };
    // the bank stream starts before anything else: the first tiles land while the queries are loaded and split
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t);

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
        label[mt] = a.labels[min(qr, a.B - 1)];      // used behind wait_vm0() below: no wait in front of the query loads
        if constexpr (!FP8) {
            // parked in AGPRs once (an asm-visible AGPR definition; without it the values stay in VGPRs and every use
            // pays four v_accvgpr_write); rows beyond the batch re-read the last query - their statistics are never stored
            const bf16_t* qp = a.q + (size_t)min(qr, a.B - 1) * a.ldq + (lane >> 4) * 8;
            // all loads of this half first, then the parking (a park right behind its load serialises 48 round trips)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[mt][ks] = *(const bf16x8*)(qp + ks * 32);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (q_parked((mt * KS + ks) * 4, 4)) asm volatile("" : "+a"(qf[mt][ks]));
                else asm volatile("" : "+v"(qf[mt][ks]));
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
                if (q_parked((mt * KS + ks) * 4, 4)) asm volatile("" : "+a"(qh[mt][ks]), "+a"(ql[mt][ks]));   // park both terms
                else asm volatile("" : "+v"(qh[mt][ks]), "+v"(ql[mt][ks]));
            }
        }
    }
    if constexpr (FP8) {
        for (int i = lane; i < r_hi - r_lo; i += 64) Ssc[i] = a.bank_scale[r_lo + i];
    }
    wait_vm0();                                       // queries, scales and the first tiles are in
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) label[mt] = q_ok[mt] ? label[mt] - (int64_t)a.m_begin : -1;

    float st_m[2] = {-INFINITY, -INFINITY}, st_l[2] = {0.f, 0.f}, st_sl[2] = {0.f, 0.f}, st_lab[2] = {-INFINITY, -INFINITY};
    const int arow = lane & 7, acol = lane >> 4;      // fragment row (lanes 8-15 repeat rows 0-7), lane quarter
    const int aswz = XG == 16 ? 2 * arow : arow;
    for (int t = 0; t < ntiles; ++t) {
        // the slot refilled now was read one tile ago: its fragment reads have returned (their MFMAs were issued)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue(t + NS - 1);
        wait_vmcnt<(NS - 1) * NP>();                  // tile t has landed; NS - 1 newer tiles stay in flight
        __builtin_amdgcn_sched_barrier(0);
        if (gm.dbg & 1) continue;
        const char* T = ring + (t % NS) * TILE_B + arow * ROWB;
        f32x4 s[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        [[maybe_unused]] f32x4 s2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};      // e4m3: the low query term
        // Software pipeline over groups of four k-steps: the fragment reads of group g + 1 are issued in front of the MFMAs
        // of group g (one wave per SIMD: a read -> wait -> MFMA chain per k-step left the wave latency-bound at ~150 cycles
        // per step, 4.4 TB/s over the chip).  The MFMAs are inline asm (queries and accumulators in AGPRs), which also
        // keeps them in program order behind the reads.
        constexpr int G = 4, NG = KS / G;
        static_assert(KS % G == 0, "k-steps per group");
        // the accumulators were just zeroed by v_accvgpr_write: VALU write -> XDL SrcC read wants wait states the hazard
        // recogniser cannot place in front of inline asm
        if constexpr (FP8) asm volatile("s_nop 4" : "+a"(s[0]), "+a"(s[1]), "+a"(s2[0]), "+a"(s2[1]));
        else asm volatile("s_nop 4" : "+a"(s[0]), "+a"(s[1]));
        if constexpr (FP8) {
            long af[2][G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int u = j * 4 + acol;
                af[0][j] = *(const long*)(T + (((u >> 1) ^ aswz) << 4) + (u & 1) * 8);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        const int u = ((g + 1) * G + j) * 4 + acol;
                        af[(g + 1) & 1][j] = *(const long*)(T + (((u >> 1) ^ aswz) << 4) + (u & 1) * 8);
                    }
                }
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        if (q_parked((mt * KS + g * G + j) * 4, 4)) {
                            mfma16_fp8_aq(s[mt], af[g & 1][j], qh[mt][g * G + j]);
                            mfma16_fp8_aq(s2[mt], af[g & 1][j], ql[mt][g * G + j]);
                        } else {
                            mfma16_fp8_vq(s[mt], af[g & 1][j], qh[mt][g * G + j]);
                            mfma16_fp8_vq(s2[mt], af[g & 1][j], ql[mt][g * G + j]);
                        }
                    }
            }
            mfma_drain(s[0], s[1], s2[0], s2[1]);
        } else {
            bf16x8 af[2][G];
#pragma unroll
            for (int j = 0; j < G; ++j) af[0][j] = *(const bf16x8*)(T + (((j * 4 + acol) ^ aswz) << 4));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        af[(g + 1) & 1][j] = *(const bf16x8*)(T + (((((g + 1) * G + j) * 4 + acol) ^ aswz) << 4));
                }
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        if (q_parked((mt * KS + g * G + j) * 4, 4)) mfma16_aq(s[mt], af[g & 1][j], qf[mt][g * G + j]);
                        else mfma16_vq(s[mt], af[g & 1][j], qf[mt][g * G + j]);
                    }
            }
            mfma_drain(s[0], s[1]);
        }
        // lane quarters 0 / 1: query (lane & 15) of each mt, keys key0 .. key0 + 3 (rows 0-3 / 4-7 of the tile); quarters
        // 2 / 3 hold the duplicate rows
        const int key0 = r_lo + t * 8 + acol * 4;
        const bool mine = acol < 2 && key0 < r_hi;
        f32x4 sc4 = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FP8) sc4 = *(const f32x4*)(Ssc + t * 8 + (acol & 1) * 4);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 v;
            float tm = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = mine && key0 + r < r_hi;
                float acc = s[mt][r];
                if constexpr (FP8) acc = (acc + s2[mt][r] * 0.0625f) * sq[mt];
                v[r] = live ? acc * sc4[r] * a.inv_tau : -INFINITY;
                tm = fmaxf(tm, v[r]);
            }
            if (tm > -INFINITY && !(gm.dbg & 4)) {
                const float mn = fmaxf(st_m[mt], tm);
                float add = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = v[r] > -INFINITY;
                    add += live ? __expf(v[r] - mn) : 0.f;
                    st_sl[mt] += live ? v[r] : 0.f;
                    st_lab[mt] = (live && (int64_t)(key0 + r) == label[mt]) ? v[r] : st_lab[mt];
                }
                st_l[mt] = st_l[mt] * __expf(st_m[mt] - mn) + add;
                st_m[mt] = mn;
            }
            if constexpr (SAVE) {
                // ldz is a multiple of 32: the vector store stays inside the row; keys beyond the shard hold -inf
                if (mine && q_ok[mt] && !(gm.dbg & 2)) *(f32x4*)(zsave + (size_t)(q0 + mt * 16 + (lane & 15)) * ldz + key0) = v;
            }
        }
    }
    // merge the key groups of a query (lanes l, l^16, l^32, l^48), then the four waves through LDS
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
    wait_vm0();                                       // the zero-fill tail tiles still target this wave's ring
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
    const size_t lds = 4 * (size_t)RT_RING_B + 4 * RQ * 4 * sizeof(float) + (FP8 ? 4 * RT_MAX_ROWS_PER_WAVE * sizeof(float) : 0);
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
    int sliced;      // the bank pointer holds the slice-major copy [D/128][M][128] (bank_slice_major)
    int dbg;         // SPN_BANK2_DBG (experiments; results wrong): 8 = stream only, 16 = no G (exp) computation
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
    g.sliced = 0;
    static const int dbg = [] { const char* e = spn_env("SPN_BANK2_DBG"); return e ? atoi(e) : 0; }();
    g.dbg = dbg;
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
                off = gm.sliced ? ((uint32_t)sl * (uint32_t)a.M + (uint32_t)(mrow0 + r)) * 128u + (uint32_t)((lane & 7) * 16)
                                : (uint32_t)(mrow0 + r) * (uint32_t)ROWB + (uint32_t)(sl * 128 + (lane & 7) * 16);
            } else {                                       // 4 rows x 256 B per instruction, swizzled image
                const int r = ii * 4 + (lane >> 4);
                const int c = (lane & 15) ^ bank2_swz(r & 15);
                off = gm.sliced ? ((uint32_t)sl * (uint32_t)a.M + (uint32_t)(mrow0 + r)) * 256u + (uint32_t)(c * 16)
                                : (uint32_t)(mrow0 + r) * (uint32_t)ROWB + (uint32_t)(sl * 256 + c * 16);
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

    issue(0);                                              // the stream starts before anything else
    issue(1);
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
    for (int i = 0; i < nmine; ++i) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot refilled now was read one tile ago
        __builtin_amdgcn_sched_barrier(0);
        issue(i + 2);
        wait_vmcnt<2 * NDMA>();
        __builtin_amdgcn_sched_barrier(0);
        if (gm.dbg & 8) continue;
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
                float gv = ((gm.dbg & 16) ? z : __expf(z - lse[mt])) - label_smoothing * inv_m_total;
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
// Default OFF (measured, LABNOTES.md section 7.6): at B = 32, M = 40 000 the pair needs 23.7 / 26.9 us against the
// first-generation kernels' 19.7 / 23.1 us - the per-wave serial structure (load queries, wait, stream) has more fixed
// cost than the block-cooperative kernels, and at that bank size fixed cost is half of the pass.  SPN_BANK2=1 or
// spn_bank_config(1) selects them (they win on the backward pass of large e4m3 banks: 137 vs 151 us at 400 000 rows).
// spn_bank_config mode: 0 = default routing, 1 = these kernels below 128 queries, 2 = the fused single pass also at
// B >= 256, 3 = two passes everywhere (no fused pass), 4 = default routing with the e4m3 fused pass on the kernel that keeps a
// bf16 tile image (the fallback of very long chunks).  Environment defaults: SPN_BANK2=1 -> 1, SPN_BANK_FUSED_LARGE=1 -> 2,
// SPN_BANK_FUSED=0 -> 3.
static bool bank2_on() { return bank_mode() == 1; }   // bank.hip: spn_bank_config / SPN_BANK2

// The saved-logits pair serves plain (one row = one target) banks, bf16 or e4m3, at per-call batches below 128 queries
bool bank_saved_path(const BankArgs& a) {
    if (!bank2_on() || a.group || a.B >= 128 || a.B <= 0) return false;
    switch (a.D) {
        case 128: case 256: case 512: case 640: case 768: case 1024: return true;
        default: return false;
    }
}

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
    DSliceGeom g = dslice_geom(a.B, a.M, a.D);
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
