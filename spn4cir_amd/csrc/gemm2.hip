// Tile-templated bf16 MFMA GEMMs (gfx950), second generation.
//
//   gemm_nt2 : C[M,N]  = epilogue(A[M,K] . B[N,K]^T)
//   gemm_tn2 : C[N1,N2] = A[Kr,N1]^T . B[Kr,N2]  (+ column sums of A = bias gradient from an
//              all-ones MFMA operand)
//
// Template <BM, BN, WM, WN, STAGES>: BM x BN output tile per workgroup of WM x WN waves, each wave
// a (BM/WM) x (BN/WN) sub-tile built from v_mfma_f32_32x32x16_bf16, 64-deep K step, STAGES LDS
// buffers filled by buffer_load...lds.  With STAGES = 3 the wait is a COUNTED s_waitcnt vmcnt so one
// whole tile stays in flight across the (raw) s_barrier.  Why bigger tiles: a 128x128x64 step moves
// 32 KB of operands per 512 MFMA cycles per CU, which is about the L2->LDS bandwidth; 256x256x64
// moves 64 KB per 2048 MFMA cycles.
// The v1 kernels (gemm.hip: 128x128, 16x16x32 MFMA, 2 blocks/CU) remain selectable: SPN_GEMM_CFG=0.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#ifdef SPN_EXPERIMENTS
#include "gemm3_nt_clobbers.inc"
#endif

// in-kernel probes (clock / phase timers, bottleneck-elimination switches; several overwrite output bytes) exist only in
// -DSPN_GEMM_PROBES experiment builds: in the shipped library the tests fold to constants and the code is gone
#ifdef SPN_GEMM_PROBES
#define SPN_DBG(ep_) ((ep_).dbg)
#else
#define SPN_DBG(ep_) 0
#endif

namespace spn {

static constexpr int BK2 = 64;


__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// Accumulator staging of the epilogues: row r of the fp32 [rows][BN] LDS image keeps its 16-byte unit `unit` (4 columns)
// at slot ((unit >> 1) | (unit & 1) * BN/8) ^ (r & 15): even units in the first half of the row, odd units in the second,
// XOR-rotated by the row.  The writers (16 lanes = 16 rows, one unit) then hit 16 different slots, and so do the readers
// (16 lanes = one row, units 2u resp. 2u+1) - with the plain unit ^ (r & 15) layout the readers' stride-2 units collided
// pairwise (1 024 conflict cycles per 256x256 tile in the SQ counters).
template <int BN>
__device__ __forceinline__ int stage_slot(int unit, int r) {
    return ((unit >> 1) | ((unit & 1) * (BN / 8))) ^ (r & 15);
}

// ----------------------------------------------------------------------------------- NT
// LDS image of a [rows][64 k] bf16 tile: row r at byte r*128, logical 16-B k-chunk c at
// position c ^ ((r>>1)&7) (conflict-free for the 32-row ds_read_b128 fragments).
// (BKT = 32: rows of 64 B, 4 chunks, position c ^ ((r>>2)&3) - 4 rows share a 256-B bank row.)
template <int BKT>
__device__ __forceinline__ int nt2_swz(int r, int c) {
    return BKT == 64 ? (c ^ ((r >> 1) & 7)) : (c ^ ((r >> 2) & 3));
}

template <int PER_WAVE, int BKT>
__device__ __forceinline__ void nt2_stage(__amdgpu_buffer_rsrc_t rs, char* sT, int row0, int ld, int k0, int wid,
                                          int lane) {
    constexpr int CPR = BKT / 8, RPI = 64 / CPR, ROWB = BKT * 2;   // chunks per row, rows per DMA instruction
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int R0 = (wid * PER_WAVE + i) * RPI;
        const int r = R0 + lane / CPR;
        const int c = nt2_swz<BKT>(r, lane % CPR);
        glds16(rs, sT + R0 * ROWB, ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(k0 + c * 8)) * 2u);
    }
}

// DMA instruction i (of PER_WAVE) of this wave only: lets the main loop spread a stage over its k steps
template <int PER_WAVE, int BKT>
__device__ __forceinline__ void nt2_stage_one(__amdgpu_buffer_rsrc_t rs, char* sT, int row0, int ld, int k0, int wid,
                                              int lane, int i, int noswz = 0) {
    constexpr int CPR = BKT / 8, RPI = 64 / CPR, ROWB = BKT * 2;
    const int R0 = (wid * PER_WAVE + i) * RPI;
    const int r = R0 + lane / CPR;
    const int c = noswz ? lane % CPR : nt2_swz<BKT>(r, lane % CPR);
    glds16(rs, sT + R0 * ROWB, ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(k0 + c * 8)) * 2u);
}

template <int BKT>
__device__ __forceinline__ bf16x8 nt2_frag(const char* sT, int r, int c) {
    return *(const bf16x8*)(sT + r * (BKT * 2) + (nt2_swz<BKT>(r, c) << 4));
}

// Fast path of nt_epilogue for FULL tiles (every row / column inside the matrix, 16-byte aligned rows): the same staging,
// but straight-line - no per-row bounds, width or output-pointer branches (the general path compiles to ~470 branches,
// each closing the scheduling window of a wave).  OUT_F32 / OUT_BF16 / AUX_OUT are the (block-uniform) outputs in use.
template <int BM, int BN, int WM, int WN, int LDS_BYTES, int MODE, int ACT, bool OUT_F32, bool OUT_BF16, bool AUX_OUT>
__device__ __forceinline__ void nt_epilogue_full(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], char* smem, int m0, int n0, int wr,
                                                 int wc, int wid, int lane, const GemmEpilogue& ep) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NJ = TN / 32;
    constexpr int CR0 = LDS_BYTES / (BN * 4);
    constexpr int CHUNK = CR0 >= BM ? BM : (CR0 / 32) * 32;
    constexpr int NCH = BM / CHUNK;
    constexpr int LPR = BN / 8, RPI = 64 / LPR;
    constexpr int ITERS = CHUNK / (NW * RPI);
    static_assert(BM % CHUNK == 0 && CHUNK % (NW * RPI) == 0 && CHUNK % 32 == 0, "full-tile epilogue geometry");
    float* sC = (float*)smem;
#ifdef SPN_GEMM_PROBES
    // SPN_GEMM_DBG bit 128: cycle stamps of the epilogue's sub-phases (mid-grid block, wave 0), over output bytes 16..
    uint32_t stamp[12];
    int ns = 0;
    const bool probe = (SPN_DBG(ep) & 128) && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
    const uint64_t pt0 = __builtin_readcyclecounter();
#define SPN_EPI_STAMP() do { if (ns < 12) stamp[ns++] = (uint32_t)(__builtin_readcyclecounter() - pt0); } while (0)
#else
#define SPN_EPI_STAMP() do { } while (0)
#endif
    const int u = lane % LPR, n = n0 + u * 8;
    f32x4 bias_lo = {0, 0, 0, 0}, bias_hi = {0, 0, 0, 0};
    if (ep.bias) {
        bias_lo = *(const f32x4*)(ep.bias + n);
        bias_hi = *(const f32x4*)(ep.bias + n + 4);
    }
    const float alpha = ep.alpha;
    const int row_l = wid * RPI + lane / LPR;                     // first row of this lane inside a chunk
    // write-through stores (ep.store_wt, block-uniform): `buffer_store_dwordx4 ... sc1` through a descriptor over the whole
    // output; the line is not kept in the XCD's L2 (MI355X_MICROARCH.md, stores of each flavour)
    const bool wt = ep.store_wt != 0;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_o16, rs_o32, rs_aux;
    if (wt) {
        if constexpr (OUT_BF16) rs_o16 = make_rsrc(ep.out_bf16, 0xfffffff0u);
        if constexpr (OUT_F32) rs_o32 = make_rsrc(ep.out_f32, 0xfffffff0u);
        if constexpr (AUX_OUT) rs_aux = make_rsrc(ep.aux_out, 0xfffffff0u);
    }
    auto put16 = [&](void* base, const __amdgpu_buffer_rsrc_t& rs, size_t byte_off, u32x4 v) {
        if (wt) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (uint32_t)byte_off, 0, 16);
        else *(u32x4*)((char*)base + byte_off) = v;
    };
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const size_t obase = (size_t)(m0 + ch * CHUNK + row_l) * ep.ldc + n;
        [[maybe_unused]] bf16x8 pf_aux[ITERS];
        [[maybe_unused]] f32x4 pf_r0[ITERS], pf_r1[ITERS];
        if constexpr (MODE == GEMM_DACT) {
            if (ep.aux_ld) {                        // block-uniform: streaming loads (read once, no reuse: keep them out of the L2's way)
                const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(ep.aux_in, 0xfffffff0u);
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const uint32_t off = (uint32_t)((obase + (size_t)it * NW * RPI * ep.ldc) * 2);
                    u32x4 v;
                    if (ep.aux_ld == 2) v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 2);
                    else if (ep.aux_ld == 16) v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 16);
                    else v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 17);
                    pf_aux[it] = __builtin_bit_cast(bf16x8, v);
                }
            } else {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) pf_aux[it] = *(const bf16x8*)(ep.aux_in + obase + (size_t)it * NW * RPI * ep.ldc);
            }
        } else if constexpr (MODE == GEMM_RESID) {
            const float* rp = ep.resid + (size_t)(m0 + ch * CHUNK + row_l) * ep.ldr + n;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                pf_r0[it] = *(const f32x4*)(rp + (size_t)it * NW * RPI * ep.ldr);
                pf_r1[it] = *(const f32x4*)(rp + (size_t)it * NW * RPI * ep.ldr + 4);
            }
        }
        __syncthreads();   // operand tiles (or the previous chunk) are no longer read
        SPN_EPI_STAMP();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if ((wr * TM + i * 32) / CHUNK != ch) continue;       // wave-uniform; compile-time when TM <= CHUNK
            const int r = wr * TM + i * 32 + (lane & 31) - ch * CHUNK;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int unit = (wc * TN + j * 32 + 8 * g) / 4 + (lane >> 5);
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *(f32x4*)(sC + r * BN + (stage_slot<BN>(unit, r) << 2)) = v;
                }
        }
        SPN_EPI_STAMP();
        __syncthreads();
        SPN_EPI_STAMP();
        // read-back in batches of RB rows: all LDS reads of a batch in flight, then its math and stores
        constexpr int RB = ITERS >= 4 ? 4 : ITERS;
        static_assert(ITERS % RB == 0, "read-back batches");
#pragma unroll
        for (int it0 = 0; it0 < ITERS; it0 += RB) {
        f32x4 t0[RB], t1[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int rr = row_l + (it0 + q) * NW * RPI;
            t0[q] = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u, rr) << 2));
            t1[q] = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u + 1, rr) << 2));
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int it = it0 + q;
            f32x4 v0 = t0[q] * alpha + bias_lo, v1 = t1[q] * alpha + bias_hi;
            const size_t o = obase + (size_t)it * NW * RPI * ep.ldc;
            auto pack8 = [](f32x4 x, f32x4 y) {
                bf16x8 p = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
                return p;
            };
            if constexpr (MODE == GEMM_STORE && ACT != ACT_NONE) {
                if constexpr (AUX_OUT) {
                    if (ep.aux_grad) {                            // block-uniform
                        f32x4 g0, g1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            act_and_grad_into(ACT, v0[e], v0[e], g0[e]);
                            act_and_grad_into(ACT, v1[e], v1[e], g1[e]);
                        }
                        put16(ep.aux_out, rs_aux, o * 2, __builtin_bit_cast(u32x4, pack8(g0, g1)));
                    } else {
                        put16(ep.aux_out, rs_aux, o * 2, __builtin_bit_cast(u32x4, pack8(v0, v1)));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v0[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v0[e]) : gelu_erf_f(v0[e]);
                            v1[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v1[e]) : gelu_erf_f(v1[e]);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v0[e]) : gelu_erf_f(v0[e]);
                        v1[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v1[e]) : gelu_erf_f(v1[e]);
                    }
                }
            } else if constexpr (MODE == GEMM_RESID) {
                v0 += pf_r0[it];
                v1 += pf_r1[it];
            } else if constexpr (MODE == GEMM_DACT) {
                const bf16x8 p = pf_aux[it];
                if (ep.aux_grad) {                                // block-uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] *= bf2f(p[e]); v1[e] *= bf2f(p[4 + e]); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = bf2f(p[e]), x1 = bf2f(p[4 + e]);
                        v0[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x0) : gelu_erf_grad_f(x0);
                        v1[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x1) : gelu_erf_grad_f(x1);
                    }
                }
            }
            if constexpr (OUT_F32) {
                put16(ep.out_f32, rs_o32, o * 4, __builtin_bit_cast(u32x4, v0));
                put16(ep.out_f32, rs_o32, o * 4 + 16, __builtin_bit_cast(u32x4, v1));
            }
            if constexpr (OUT_BF16) put16(ep.out_bf16, rs_o16, o * 2, __builtin_bit_cast(u32x4, pack8(v0, v1)));
        }
        }
        SPN_EPI_STAMP();
    }
#ifdef SPN_GEMM_PROBES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPN_EPI_STAMP();
    if (probe) {
        uint32_t* o = ep.out_f32 ? (uint32_t*)ep.out_f32 : (uint32_t*)ep.out_bf16;
        for (int q = 0; q < 12; ++q) o[4 + q] = q < ns ? stamp[q] : 0u;
    }
#endif
#undef SPN_EPI_STAMP
}

// Epilogue shared by the NT kernels.  (B-frag, A-frag) operand order: lane owns row m = lane&31 and, for g = 0..3,
// the 4 consecutive columns n = 8g + 4*(lane>>5) + 0..3 of each 32x32 tile (regs 4g..4g+3).
// Stores straight from that layout touch 16 B per row per instruction, so the accumulators are
// first staged through LDS (fp32, 16-B units XOR-swizzled by row) and the epilogue math + all
// global traffic run row-major: one wave instruction = one or two whole rows, fully coalesced.
template <int BM, int BN, int WM, int WN, int LDS_BYTES, int MODE, int ACT>
__device__ __forceinline__ void nt_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], char* smem, int M, int N, int m0,
                                            int n0, int wr, int wc, int wid, int lane, const GemmEpilogue& ep) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NJ = TN / 32;
    if constexpr (MODE != GEMM_BANKSTATS && BM == 256 && BN == 256 && LDS_BYTES >= 131072) {
        // full tile with 16-byte aligned rows and one of the usual output combinations: the straight-line path
        const bool full = !ep.direct_store && m0 + BM <= M && n0 + BN <= N && ep.ldc % 8 == 0 &&
                          (MODE != GEMM_RESID || ep.ldr % 4 == 0) && !(SPN_DBG(ep) & 256);
        if (full) {
            const bool f32o = ep.out_f32 != nullptr, b16o = ep.out_bf16 != nullptr, aux = ep.aux_out != nullptr;
#define SPN_EPI_FULL(F_, B_, A_) \
            nt_epilogue_full<BM, BN, WM, WN, LDS_BYTES, MODE, ACT, F_, B_, A_>(acc, smem, m0, n0, wr, wc, wid, lane, ep)
            if constexpr (MODE == GEMM_STORE && ACT != ACT_NONE) {
                if (b16o && !f32o && aux) { SPN_EPI_FULL(false, true, true); return; }
                if (b16o && !f32o && !aux) { SPN_EPI_FULL(false, true, false); return; }
            } else {
                if (b16o && !f32o) { SPN_EPI_FULL(false, true, false); return; }
                if (f32o && !b16o) { SPN_EPI_FULL(true, false, false); return; }
            }
#undef SPN_EPI_FULL
        }
    }
    if (!ep.direct_store) {
        constexpr int CR0 = LDS_BYTES / (BN * 4);
        constexpr int CHUNK = CR0 >= BM ? BM : (CR0 / 32) * 32;     // rows per staging pass
        constexpr int NCH = (BM + CHUNK - 1) / CHUNK;
        // a lane owns 8 consecutive columns of a row: bf16 traffic moves 16 B per lane (the store path is
        // issue-bound, so half as many, twice as wide instructions), fp32 traffic as two 16-B accesses
        constexpr int LPR = BN / 8, RPI = 64 / LPR;                 // lanes per row, rows per wave instruction
        static_assert(LPR <= 64 && 64 % LPR == 0, "row mapping");
        float* sC = (float*)smem;
        const int u = lane % LPR, n = n0 + u * 8;
        const bool hi = n + 4 < N;                                  // N % 4 == 0: each half is all in or all out
        const bool wide = hi && (ep.ldc % 8 == 0);
        f32x4 bias_lo = {0, 0, 0, 0}, bias_hi = {0, 0, 0, 0};
        if (ep.bias && n < N) bias_lo = *(const f32x4*)(ep.bias + n);
        if (ep.bias && hi) bias_hi = *(const f32x4*)(ep.bias + n + 4);
        constexpr int ITERS = (CHUNK + NW * RPI - 1) / (NW * RPI);   // rows of a chunk handled by one lane
        for (int ch = 0; ch < NCH; ++ch) {
            // the epilogue's global READS of this chunk (pre-activation / residual) are issued first, so that their
            // latency hides behind the LDS staging of the accumulators instead of stalling every row
            [[maybe_unused]] bf16x8 pf_aux[ITERS];
            [[maybe_unused]] f32x4 pf_r0[ITERS], pf_r1[ITERS];
            [[maybe_unused]] int64_t bs_lab[ITERS];            // BANKSTATS: the rows' labels, fetched ahead of the staging
            if constexpr (MODE == GEMM_BANKSTATS) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int rr = wid * RPI + lane / LPR + it * NW * RPI;
                    const int m = min(m0 + ch * CHUNK + rr, M - 1);
                    bs_lab[it] = ep.bs_labels[m] - (int64_t)ep.bs_m_begin;
                }
            }
            if constexpr (MODE == GEMM_DACT || MODE == GEMM_RESID) {
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int rr = wid * RPI + lane / LPR + it * NW * RPI;
                    const int m = m0 + ch * CHUNK + rr;
                    const bool ok = rr < CHUNK && m < M && n < N;
                    if constexpr (MODE == GEMM_DACT) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf_aux[it][e] = (bf16_t)0.f;
                        if (ok) {
                            const size_t o = (size_t)m * ep.ldc + n;
                            if (wide) {
                                pf_aux[it] = *(const bf16x8*)(ep.aux_in + o);
                            } else {
                                const bf16x4 p0 = *(const bf16x4*)(ep.aux_in + o);
                                bf16x4 p1 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                                if (hi) p1 = *(const bf16x4*)(ep.aux_in + o + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { pf_aux[it][e] = p0[e]; pf_aux[it][4 + e] = p1[e]; }
                            }
                        }
                    } else {
                        pf_r0[it] = f32x4{0, 0, 0, 0};
                        pf_r1[it] = f32x4{0, 0, 0, 0};
                        if (ok) {
                            const float* rp = ep.resid + (size_t)m * ep.ldr + n;
                            pf_r0[it] = *(const f32x4*)rp;
                            if (hi) pf_r1[it] = *(const f32x4*)(rp + 4);
                        }
                    }
                }
            }
            __syncthreads();   // operand tiles (or the previous chunk) are no longer read
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rl = wr * TM + i * 32 + (lane & 31);
                if (rl / CHUNK != ch) continue;
                const int r = rl - ch * CHUNK;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int unit = (wc * TN + j * 32 + 8 * g) / 4 + (lane >> 5);
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *(f32x4*)(sC + r * BN + (stage_slot<BN>(unit, r) << 2)) = v;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int rr = wid * RPI + lane / LPR + it * NW * RPI;
                const int m = m0 + ch * CHUNK + rr;
                if (rr >= CHUNK || m >= M) continue;          // wave-uniform per row pair for the shuffles below
                if (MODE != GEMM_BANKSTATS && n >= N) continue;
                f32x4 v0 = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u, rr) << 2));
                f32x4 v1 = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u + 1, rr) << 2));
                if constexpr (MODE == GEMM_BANKSTATS) {
                    // this lane's 8 logits of row m -> {max, sum exp, sum, label logit}, merged over the LPR lanes
                    // that share the row (consecutive lanes: xor shuffles below LPR stay inside the row)
                    const int64_t lab = bs_lab[it];
                    float mx = -INFINITY, sl = 0.f, lv = -INFINITY, l = 0.f;
                    float z[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {          // predicates, not branches (one exec-mask branch per logit otherwise)
                        const float x = (e < 4 ? v0[e] : v1[e - 4]) * ep.bs_inv_tau;
                        const bool ok = n + e < N;
                        z[e] = ok ? x : -INFINITY;
                        mx = fmaxf(mx, z[e]);
                        sl += ok ? x : 0.f;
                        lv = (ok & ((int64_t)(n + e) == lab)) ? x : lv;
                    }
                    // row maximum over the tile first (shuffles only), then ONE exponential per logit against it: the
                    // pairwise online merge cost two more exponentials per lane and shuffle stage (18 instead of 8)
#pragma unroll
                    for (int off = LPR / 2; off > 0; off >>= 1) {
                        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                        sl += __shfl_xor(sl, off, 64);
                        lv = fmaxf(lv, __shfl_xor(lv, off, 64));
                    }
                    if (mx > -INFINITY) {
                        float pe[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { pe[e] = __expf(z[e] - mx); l += pe[e]; }
                        if (ep.bs_p_out) {                 // block-uniform: tile-relative probabilities for the backward pass
                            const bf16x8 pk = {f2bf(pe[0]), f2bf(pe[1]), f2bf(pe[2]), f2bf(pe[3]),
                                               f2bf(pe[4]), f2bf(pe[5]), f2bf(pe[6]), f2bf(pe[7])};
                            *(bf16x8*)(ep.bs_p_out + (size_t)m * ep.bs_ldp + n) = pk;      // bs_ldp covers whole tiles
                        }
                    }
#pragma unroll
                    for (int off = LPR / 2; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
                    if (u == 0) {
                        const int tile_n = n0 / BN;
                        *(f32x4*)(ep.bs_out + ((size_t)tile_n * M + m) * 4) = f32x4{mx, l, sl, lv};
                        if (ep.bs_max_out) ep.bs_max_out[(size_t)tile_n * M + m] = mx;
                    }
                    continue;
                }
                v0 = v0 * ep.alpha + bias_lo;
                v1 = v1 * ep.alpha + bias_hi;
                const size_t o = (size_t)m * ep.ldc + n;
                auto pack8 = [](f32x4 x, f32x4 y) {
                    bf16x8 p = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
                    return p;
                };
                auto store_bf16 = [&](bf16_t* dst, f32x4 x, f32x4 y) {
                    if (wide) {
                        *(bf16x8*)(dst + o) = pack8(x, y);
                    } else {
                        bf16x4 p = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3])};
                        *(bf16x4*)(dst + o) = p;
                        if (hi) {
                            bf16x4 q = {f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
                            *(bf16x4*)(dst + o + 4) = q;
                        }
                    }
                };
                if constexpr (MODE == GEMM_STORE) {
                    if constexpr (ACT != ACT_NONE) {
                        if (ep.aux_out && ep.aux_grad) {       // wave-uniform: keep act'(pre) instead of pre
                            f32x4 g0, g1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                act_and_grad_into(ACT, v0[e], v0[e], g0[e]);
                                act_and_grad_into(ACT, v1[e], v1[e], g1[e]);
                            }
                            store_bf16(ep.aux_out, g0, g1);
                        } else {
                            if (ep.aux_out) store_bf16(ep.aux_out, v0, v1);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v0[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v0[e]) : gelu_erf_f(v0[e]);
                                v1[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v1[e]) : gelu_erf_f(v1[e]);
                            }
                        }
                    }
                } else if constexpr (MODE == GEMM_RESID) {
                    v0 += pf_r0[it];
                    v1 += pf_r1[it];
                } else if constexpr (MODE == GEMM_DACT) {
                    const bf16x8 p = pf_aux[it];
                    if (ep.aux_grad) {                         // wave-uniform: aux_in already holds act'(pre)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] *= bf2f(p[e]); v1[e] *= bf2f(p[4 + e]); }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x0 = bf2f(p[e]), x1 = bf2f(p[4 + e]);
                            v0[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x0) : gelu_erf_grad_f(x0);
                            v1[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x1) : gelu_erf_grad_f(x1);
                        }
                    }
                }
                if (ep.out_f32) {
                    *(f32x4*)(ep.out_f32 + o) = v0;
                    if (hi) *(f32x4*)(ep.out_f32 + o + 4) = v1;
                }
                if (ep.out_bf16) store_bf16(ep.out_bf16, v0, v1);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wr * TM + i * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * TN + j * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= N) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                v *= ep.alpha;
                if (ep.bias) v += *(const f32x4*)(ep.bias + n);
                const size_t o = (size_t)m * ep.ldc + n;
                if constexpr (MODE == GEMM_STORE) {
                    if constexpr (ACT != ACT_NONE) {
                        f32x4 a = v;
                        if (ep.aux_out && ep.aux_grad) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) act_and_grad_into(ACT, v[e], v[e], a[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v[e]) : gelu_erf_f(v[e]);
                        }
                        if (ep.aux_out) {
                            bf16x4 p = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
                            *(bf16x4*)(ep.aux_out + o) = p;
                        }
                    }
                } else if constexpr (MODE == GEMM_RESID) {
                    v += *(const f32x4*)(ep.resid + (size_t)m * ep.ldr + n);
                } else if constexpr (MODE == GEMM_DACT) {
                    const bf16x4 p = *(const bf16x4*)(ep.aux_in + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = bf2f(p[e]);
                        v[e] *= ep.aux_grad ? x : (ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x) : gelu_erf_grad_f(x));
                    }
                }
                if (ep.out_f32) *(f32x4*)(ep.out_f32 + o) = v;
                if (ep.out_bf16) {
                    bf16x4 p = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    *(bf16x4*)(ep.out_bf16 + o) = p;
                }
            }
        }
    }
}

// SPN_EXP_NOBAR (experiment builds, tools/build_variant.sh; results are racy): 1 = no barrier behind the MFMA section,
// 2 = no barrier at all - prices the eight barriers per k tile
#ifndef SPN_EXP_NOBAR
#define SPN_EXP_NOBAR 0
#endif
// SPN_EXP_XBAR=1 (experiment): every slot barrier twice - prices one barrier event of the 8-wave workgroup
#ifndef SPN_EXP_XBAR
#define SPN_EXP_XBAR 0
#endif
#define SPN_SLOT_BAR(LVL) do { if (SPN_EXP_NOBAR < LVL) { __builtin_amdgcn_s_barrier(); for (int xb__ = 0; xb__ < SPN_EXP_XBAR; ++xb__) __builtin_amdgcn_s_barrier(); } } while (0)
template <int BM, int BN, int WM, int WN, int STAGES, int MODE, int ACT, int SCHED, int BKT>
__global__ __launch_bounds__(WM* WN * 64, (STAGES * (BM + BN) * BKT * 2 <= 80 * 1024 ? 2 : 1) * (WM * WN) / 4) void gemm_nt2_kernel(const bf16_t* __restrict__ A,
                                                                            const bf16_t* __restrict__ B, int M,
                                                                            int N, int K, int lda, int ldb,
                                                                            GemmEpilogue ep) {
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int KSTEPS = BKT / 16;
    constexpr int GA = A_BYTES / 1024 / NW, GB = B_BYTES / 1024 / NW;      // DMA instructions per wave per stage
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NJ = TN / 32;
    static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0 && TM % 32 == 0 && TN % 32 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // SPN_GEMM_DBG bit 64: phase times (shader cycles) of a mid-grid block - setup / prologue (first DMA + wait) /
    // k loop / epilogue - written over the first 16 bytes of the output
    const uint64_t ph0 = (SPN_DBG(ep) & 64) ? __builtin_readcyclecounter() : 0;
    uint64_t ph2 = 0;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    if (ep.stag_ticks > 0 && (int)blockIdx.x >= ep.stag_from && (int)blockIdx.x < ep.stag_to) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)ep.stag_ticks) __builtin_amdgcn_s_sleep(32);
    }
    const int tiles_n = (N + BN - 1) / BN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    int m0, n0;
    if (ep.col_group > 0 && ep.col_group < tiles_n) {
        // column-group-major tile order (launch_nt2): logical ids sweep col_group column tiles for every row block, then the
        // next group - the contiguous id range of an XCD then needs col_group B panels only, which stay in its L2
        const int tiles_m = (M + BM - 1) / BM;
        int c0 = 0, wg = ep.col_group;
        while (bid >= tiles_m * wg) {              // <= tiles_n / col_group iterations, block-uniform
            bid -= tiles_m * wg;
            c0 += wg;
            wg = min(ep.col_group, tiles_n - c0);
        }
        m0 = (bid / wg) * BM;
        n0 = (c0 + bid % wg) * BN;
    } else {
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)M * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)N * (uint32_t)ldb * 2u);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = zero16();

    // SPN_GEMM_DBG bit 32: clock probe - shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) spent in
    // the main loop of the last tile, written over the first 8 bytes of the bf16 output
    const uint64_t dbg_c0 = (SPN_DBG(ep) & (32 | 64)) ? __builtin_readcyclecounter() : 0;
    const uint64_t dbg_r0 = (SPN_DBG(ep) & 32) ? __builtin_amdgcn_s_memrealtime() : 0;

    const int nk = K / BKT;
    auto stage = [&](int kt, int buf) {
        char* s = smem + buf * STAGE;
        nt2_stage<GA, BKT>(rsA, s, m0, lda, kt * BKT, wid, lane);
        nt2_stage<GB, BKT>(rsB, s + A_BYTES, n0, ldb, kt * BKT, wid, lane);
    };
    if constexpr (SCHED == 2) {
        // ---- 8-slot staggered schedule (cdna_hip_programming.md "256^2 8-phase template", own chunking) ----
        // Fixed geometry: 256x256x64 tile, 8 waves (2 x 4), 2 LDS buffers of one k tile each.  A k tile is
        // consumed in 4 slots, one 64x32 quadrant of the wave's 128x64 output per slot:
        //   slot 0: read A rows mq0 (8 x b128) + B rows nq0 (4)  -> acc[0..1][0]
        //   slot 1: read B rows nq1 (4)                          -> acc[0..1][1]
        //   slot 2: read A rows mq1 (8)                          -> acc[2..3][1]
        //   slot 3: nothing                                      -> acc[2..3][0]
        // and staged as 4 chunks of 16 KB (2 DMA per wave each), chunk c of k tile j = the rows read in ONE slot:
        //   c0 = A mq0, c1 = B nq0 (both read in slot 0), c2 = B nq1 (slot 1), c3 = A mq1 (slot 2).
        // Every slot is  [LDS reads][one chunk of DMA][counted vmcnt] s_barrier [MFMA x8] s_barrier , and the
        // waves of the lower M half (wr == 1) run one barrier behind the upper half: on every SIMD one wave is in
        // its MFMA section while its partner issues memory instructions.  Issue order  c2(j+1), c3(j+1), c0(j+2),
        // c1(j+2)  in slots 0..3 of k tile j: a chunk is re-staged >= 2 slots after its last read, and a whole k
        // tile (4 chunks = 8 DMA per wave) stays in flight behind each wait: vmcnt(8), never 0 in steady state.
        // RAW: a chunk is read one slot after the slot whose pre-barrier wait retired it (both wave halves have
        // then executed that wait before a barrier the reader has passed).
        static_assert(BM == 256 && BN == 256 && WM == 2 && WN == 4 && STAGES == 2 && BKT == 64, "phased schedule geometry");
// SPN_NT_SPLIT_DMA=1 (experiment): the second DMA of a slot's chunk is issued inside the slot's MFMA section (after four
// MFMA) instead of in its memory section - eight DMA of the four waves of a half no longer hit the address unit as one burst
#ifndef SPN_NT_SPLIT_DMA
#define SPN_NT_SPLIT_DMA 0
#endif
        auto chunk = [&](int c, int j, int t0 = 0, int t1 = 2) {   // stage chunk c of k tile j (wave-uniform j < nk)
            char* sb = smem + (j & 1) * STAGE;
            const int k0 = j * BKT;
#pragma unroll
            for (int t = t0; t < t1; ++t) {
                const int i = wid * 2 + t;        // 16 DMA instructions x 8 rows per chunk
                int row0;
                if (c == 0 || c == 3) row0 = (i < 8 ? 0 : 128) + (c == 3 ? 64 : 0) + (i & 7) * 8;
                else row0 = (i >> 2) * 64 + (c == 2 ? 32 : 0) + (i & 3) * 8;
                const int r = row0 + (lane >> 3);
                const int cc = nt2_swz<BKT>(r, lane & 7);
                if (c == 0 || c == 3)
                    glds16(rsA, sb + row0 * 128, ((uint32_t)(m0 + r) * (uint32_t)lda + (uint32_t)(k0 + cc * 8)) * 2u);
                else
                    glds16(rsB, sb + A_BYTES + row0 * 128, ((uint32_t)(n0 + r) * (uint32_t)ldb + (uint32_t)(k0 + cc * 8)) * 2u);
            }
        };
        auto wait_tile = [&](bool full) {          // full: a whole newer k tile was issued behind the awaited chunk
            if (full) wait_vmcnt<SPN_NT_SPLIT_DMA ? 7 : 8>();
            else wait_vmcnt<0>();
        };
        auto wait_first = [&](bool full) {         // the wait in front of the loop: whole chunks were issued
            if (full) wait_vmcnt<8>();
            else wait_vmcnt<0>();
        };
        if (nk > 0) { chunk(0, 0); chunk(1, 0); chunk(2, 0); chunk(3, 0); }
        if (nk > 1) { chunk(0, 1); chunk(1, 1); }
        wait_first(nk > 1);
        __builtin_amdgcn_s_barrier();
        if (SPN_DBG(ep) & 64) ph2 = __builtin_readcyclecounter();
        if (wr == 1) SPN_SLOT_BAR(2);                  // stagger the lower half by one barrier (event)
        bf16x8 a[2][4], b0[4], b1[4];
        const int arow = wr * TM + (lane & 31), brow = wc * TN + (lane & 31), cl = lane >> 5;
        // The MFMA builtins carry no side effects, so instruction selection is free to float them across
        // s_barrier; the empty volatile asm statements tie their operands (after the first barrier) and their
        // results (before the second) to the slot.
#define SPN_SLOT_MFMA(I0, J, BREG, DCOND, DC, DJ)                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        SPN_SLOT_BAR(2);                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        asm volatile("" : "+v"(BREG[0]), "+v"(BREG[1]), "+v"(BREG[2]), "+v"(BREG[3]));                  \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                              \
            acc[I0][J] = mfma32(BREG[kk], a[0][kk], acc[I0][J]);                                        \
            acc[I0 + 1][J] = mfma32(BREG[kk], a[1][kk], acc[I0 + 1][J]);                                \
            if (SPN_NT_SPLIT_DMA && kk == 1) {                                                          \
                asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));                              \
                __builtin_amdgcn_sched_barrier(0);                                                      \
                if (DCOND) chunk(DC, DJ, 1, 2);                                                         \
                __builtin_amdgcn_sched_barrier(0);                                                      \
            }                                                                                           \
        }                                                                                               \
        asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));                                      \
        __builtin_amdgcn_s_setprio(0);                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        SPN_SLOT_BAR(1);                                                                                \
        __builtin_amdgcn_sched_barrier(0);
// SPN_GEMM_LOOP_DBG=1 builds keep the bottleneck-elimination switches of the loop (SPN_GEMM_DBG bits 2 and 4: no DMA /
// fragments read once); the shipped loop has no such branches in its slots
#ifndef SPN_GEMM_LOOP_DBG
#define SPN_GEMM_LOOP_DBG 0
#endif
        const bool dma = SPN_GEMM_LOOP_DBG ? !(SPN_DBG(ep) & 2) : true;
        for (int kt = 0; kt < nk; ++kt) {
            const char* sA = smem + (kt & 1) * STAGE;
            const char* sB = sA + A_BYTES;
            const bool n1 = kt + 1 < nk && dma, n2 = kt + 2 < nk && dma;
            const bool rd = SPN_GEMM_LOOP_DBG ? (!(SPN_DBG(ep) & 4) || kt == 0) : true;
            // slot 0
            if (rd) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) b0[kk] = nt2_frag<BKT>(sB, brow, kk * 2 + cl);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    a[0][kk] = nt2_frag<BKT>(sA, arow, kk * 2 + cl);
                    a[1][kk] = nt2_frag<BKT>(sA, arow + 32, kk * 2 + cl);
                }
            }
            if (n1) chunk(2, kt + 1, 0, SPN_NT_SPLIT_DMA ? 1 : 2);
            wait_tile(n1);                              // c2(kt) for slot 1
            SPN_SLOT_MFMA(0, 0, b0, n1, 2, kt + 1)
            // slot 1
            if (rd) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) b1[kk] = nt2_frag<BKT>(sB, brow + 32, kk * 2 + cl);
            }
            if (n1) chunk(3, kt + 1, 0, SPN_NT_SPLIT_DMA ? 1 : 2);
            wait_tile(n1);                              // c3(kt) for slot 2
            SPN_SLOT_MFMA(0, 1, b1, n1, 3, kt + 1)
            // slot 2
            if (rd) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    a[0][kk] = nt2_frag<BKT>(sA, arow + 64, kk * 2 + cl);
                    a[1][kk] = nt2_frag<BKT>(sA, arow + 96, kk * 2 + cl);
                }
            }
            if (n2) chunk(0, kt + 2, 0, SPN_NT_SPLIT_DMA ? 1 : 2);
            SPN_SLOT_MFMA(2, 1, b1, n2, 0, kt + 2)
            // slot 3
            if (n2) chunk(1, kt + 2, 0, SPN_NT_SPLIT_DMA ? 1 : 2);
            if (kt + 1 < nk) wait_tile(n2);             // c0, c1 of k tile kt+1 for its slot 0
            SPN_SLOT_MFMA(2, 0, b0, n2, 1, kt + 2)
        }
        if (wr == 0) SPN_SLOT_BAR(2);                  // re-align the two halves
#undef SPN_SLOT_MFMA
    } else {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);
    int cur = 0, fill = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the (STAGES-2) newer stages are still outstanding
        static_assert(STAGES <= 3, "the counted wait below assumes at most one newer tile in flight");
        if (STAGES >= 3 && kt + 1 < nk) wait_vmcnt<(STAGES - 2) * (GA + GB)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everyone's part of tile kt landed; everyone is done with tile kt-1
        // SCHED 1: the DMA of tile kt+STAGES-1 is spread over the four k16 steps (after each step's LDS reads,
        // before its MFMAs) instead of being issued as one burst behind the barrier (SCHED 0), where both waves
        // of a SIMD would stall the matrix pipe together.
        const bool do_stage = kt + STAGES - 1 < nk && !(SPN_DBG(ep) & 2);
        char* sF = smem + fill * STAGE;
        const int kf = (SPN_DBG(ep) & 1) ? 0 : (kt + STAGES - 1) * BKT;
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + A_BYTES;
        bf16x8 a[MI], b[NJ];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int c = kk * 2 + (lane >> 5);
            if (!(SPN_DBG(ep) & 4) || kk == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = nt2_frag<BKT>(sA, wr * TM + i * 32 + (lane & 31), c);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = nt2_frag<BKT>(sB, wc * TN + j * 32 + (lane & 31), c);
            }
            if (SCHED == 1) {
                if (do_stage) {
#pragma unroll
                    for (int q = kk * ((GA + KSTEPS - 1) / KSTEPS); q < (kk + 1) * ((GA + KSTEPS - 1) / KSTEPS) && q < GA; ++q)
                        nt2_stage_one<GA, BKT>(rsA, sF, m0, lda, kf, wid, lane, q, SPN_DBG(ep) & 16);
#pragma unroll
                    for (int q = kk * ((GB + KSTEPS - 1) / KSTEPS); q < (kk + 1) * ((GB + KSTEPS - 1) / KSTEPS) && q < GB; ++q)
                        nt2_stage_one<GB, BKT>(rsB, sF + A_BYTES, n0, ldb, kf, wid, lane, q, SPN_DBG(ep) & 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (kk == 0 && do_stage) {
                stage(kt + STAGES - 1, fill);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(b[j], a[i], acc[i][j]);
            if (SCHED == 1) __builtin_amdgcn_sched_barrier(0);
        }
        cur = cur == STAGES - 1 ? 0 : cur + 1;
        fill = fill == STAGES - 1 ? 0 : fill + 1;
    }
    }
    // Epilogue.  (B-frag, A-frag) operand order: lane owns row m = lane&31 and, for g = 0..3, the 4
    // consecutive columns n = 8g + 4*(lane>>5) + 0..3 of each 32x32 tile (regs 4g..4g+3).
    // Stores straight from that layout touch 16 B per row per instruction, so the accumulators are
    // first staged through LDS (fp32, 16-B units XOR-swizzled by row) and the epilogue math + all
    // global traffic run row-major: one wave instruction = one or two whole rows, fully coalesced.
    if ((SPN_DBG(ep) & 32) && blockIdx.x == gridDim.x - 1 && tid == 0 && ep.out_bf16) {
        uint32_t* o = (uint32_t*)ep.out_bf16;
        o[0] = (uint32_t)(__builtin_readcyclecounter() - dbg_c0);
        o[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_r0);
        return;
    }
    if (SPN_DBG(ep) & 8) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 12345.678f && ep.out_f32) ep.out_f32[0] = t;
        return;
    }
    const uint64_t ph3 = (SPN_DBG(ep) & 64) ? __builtin_readcyclecounter() : 0;
    nt_epilogue<BM, BN, WM, WN, STAGES * STAGE, MODE, ACT>(acc, smem, M, N, m0, n0, wr, wc, wid, lane, ep);
    if ((SPN_DBG(ep) & 64) && blockIdx.x == gridDim.x / 2 && tid == 0) {
        const uint64_t ph4 = __builtin_readcyclecounter();
        uint32_t* o = ep.out_f32 ? (uint32_t*)ep.out_f32 : (uint32_t*)ep.out_bf16;
        o[0] = (uint32_t)(dbg_c0 - ph0);
        o[1] = (uint32_t)(ph2 - dbg_c0);
        o[2] = (uint32_t)(ph4 - ph3);
        o[3] = (uint32_t)(ph3 - ph2);
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, int MODE, int ACT, int SCHED, int BKT>
static int launch_nt2(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, const GemmEpilogue& ep,
                      hipStream_t st) {
    constexpr int LDS = STAGES * (BM + BN) * BKT * 2;
    auto kern = gemm_nt2_kernel<BM, BN, WM, WN, STAGES, MODE, ACT, SCHED, BKT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // De-phasing the epilogue bursts (one workgroup per CU, equal tiles: all CUs store their 128-256 KB of output at the same
    // time, ~8 us per round at the HBM write rate, while the matrix pipes idle).  In a launch of several rounds whose last
    // round is partial, the CUs that will run one tile fewer have a whole tile of slack: their FIRST tile starts
    // stag_ticks later, they stay that far behind for the rest of the launch (the dispatcher hands the last round's tiles
    // to the CUs that free up first - the others) and their bursts fall into the main loops of the rest.
    GemmEpilogue e = ep;
    e.stag_from = 0;
    e.stag_ticks = 0;
    if constexpr (SCHED == 2) {
        static const int stag_us = [] { const char* v = spn_env("SPN_GEMM_STAGGER_US"); return v ? atoi(v) : 6; }();   // measured: 0 / 6 / 10 / 14 / 18 us -> 13.38 / 13.31 / 13.32 / 13.33 / 13.36 ms per step
        const int cus = device_cu_count(), rem = tiles % cus;
        if (stag_us > 0 && tiles > cus && rem > 0 && MODE != GEMM_BANKSTATS) {
            e.stag_from = rem;                                  // first-round workgroups [rem, cus) are delayed
            e.stag_to = cus;
            e.stag_ticks = stag_us * 100 * (K / 768 > 0 ? K / 768 : 1);   // s_memrealtime ticks (100 MHz), scaled with the k loop
        }
    }
    // Tile order: row-major over the column tiles by default.  With many column tiles (fc / d-activation: 12 x 393 KB of weight
    // panels = 4.7 MB) the panels of ONE row-major sweep no longer fit an XCD's 4 MB L2 next to the A panels and the output stream,
    // and every round of 32 tiles re-fetches them: FETCH_SIZE x 2 = 194 MB per fc launch against 35 MB of operands
    // (profiles/r04_bench_n1_pmc_fetch_size.txt).  In column groups of `cg` tiles an XCD's contiguous id range needs cg panels only
    // (they stay resident), at the price of the A panels being fetched once per group: A x ceil(tiles_n / cg) + 8 x cg panels.
    // cg = tiles_n split evenly into the fewest groups whose panels take <= 2.5 MB; SPN_GEMM_COL_GROUP=n forces n (0 = row-major).
    if (MODE != GEMM_BANKSTATS) {
        static const int forced = [] { const char* v = spn_env("SPN_GEMM_COL_GROUP"); return v ? atoi(v) : -1; }();
        const int tiles_n = (N + BN - 1) / BN;
        int cg = forced;
        if (cg < 0) {
            const int fit = (int)((2.5 * 1024 * 1024) / ((double)BN * K * 2));
            cg = 0;
            if (fit >= 1 && fit < tiles_n && tiles_n > 8) {
                const int groups = (tiles_n + fit - 1) / fit;
                cg = (tiles_n + groups - 1) / groups;
            }
        }
        e.col_group = cg;
        // write-through output stores for multi-round launches (their operand panels are re-read round after round): the
        // outputs must be addressable through one 32-bit buffer descriptor.  SPN_GEMM_STORE_WT=0 / 1 forces off / on.
        static const int wt = [] { const char* v = spn_env("SPN_GEMM_STORE_WT"); return v ? atoi(v) : -1; }();
        const bool fits = (uint64_t)M * (uint64_t)(ep.ldc > 0 ? ep.ldc : N) * 4 < 0xfffffff0ull;
        // measured (round 4, three paired bench runs): 13.75 -> 13.64 ms per step with every bf16 output written through; an fp32
        // output (the one RESID launch left) loses (105.7 -> 124.5 us) and keeps plain stores
        e.store_wt = (fits && (wt > 0 || (wt < 0 && !ep.out_f32))) ? 1 : 0;
        // d-activation reads 121 MB of act'(pre) exactly once: non-temporal loads (aux = 2) keep that stream from displacing the
        // operand panels (two paired runs: 13.12 / 13.07 -> 13.04 / 13.00 ms per step; sc1 / sc0 sc1: no change).  SPN_GEMM_AUX_LD=0 = plain
        static const int ald = [] { const char* v = spn_env("SPN_GEMM_AUX_LD"); return v ? atoi(v) : 2; }();
        e.aux_ld = fits ? ald : 0;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), LDS, st, A, B, M, N, K, lda, ldb, e);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

#ifdef SPN_EXPERIMENTS
// ----------------------------------------------------------------------------------- NT, persistent
// (round 4; compiled only into the experiments build, like gemm_nt3 below: bit-identical to gemm_nt2 and faster per launch on
// the 924-tile products in a rocprofv3 trace - fc + QuickGELU 124.4 -> 118.4 us - but the STEP is slower with it, 13.52 ->
// 13.66 ms in two paired bench runs: the cycles it removes are the low-power ones (a workgroup waiting for its first operands),
// and on a power-limited chip they come back as a lower clock on everything else.  profiles/r04_gemm_persist_ab.txt)
// gemm_nt2p: the 8-slot staggered schedule of gemm_nt2_kernel<.., SCHED 2> as ONE workgroup per CU that walks several output
// tiles, with the operand stream running ACROSS tile boundaries: the k tiles of a workgroup's tiles form one sequence g = 0,
// 1, 2 ... (LDS stage g & 1), and the chunk issues "k tile + 1" / "k tile + 2" of the last k tiles of a tile already fetch
// the first k tiles of the NEXT tile - when the epilogue of tile i ends, k tile 0 of tile i + 1 is in LDS (all four chunks)
// and the workgroup enters its main loop without the 2.0-2.6 k cycle prologue (first DMA + L2 / HBM latency) every
// non-persistent workgroup pays, and without a dispatch in between.  Only multi-round launches take it (more tiles than
// CUs: the qkv projection 693, fc / d-activation 924 tiles); a single-round launch has nothing to hide.
// LDS: the two 64 KB stages only.  At the end of a tile stage (g + 1) & 1 holds the next tile's k tile 0; the issue of the
// next tile's k tile 1 (chunks c0, c1, normally slots 2, 3 of the last k tile) is HELD BACK until the epilogue is done, so
// that the stage of the tile's last k tile is free for the accumulator staging: four passes of 64 rows (32 from each M half
// - all eight waves write in every pass, 8 ds_write_b128 each) instead of two passes of 128.
// vmcnt: the epilogue's loads and stores sit in the same in-order counter as the DMA; they are older than every DMA of the
// next tile, so the counted waits of the loop stay valid (they only get more conservative) - and k tile 0 of a non-first
// tile needs none: everything it reads landed before the epilogue (vmcnt(0) in slot 3 of the last k tile).
template <int MODE, int ACT, bool OUT_F32, bool OUT_BF16, bool AUX_OUT>
__device__ __forceinline__ void nt_epilogue_p(f32x16 (&acc)[4][2], char* sbuf, int m0, int n0, int wr, int wc, int wid, int lane,
                                              const GemmEpilogue& ep) {
    constexpr int BN = 256, NW = 8, LPR = 32, RPI = 2, ITERS = 4;      // 64 staging rows per pass = 4 x (8 waves x 2 rows)
    float* sC = (float*)sbuf;
    const int u = lane % LPR, n = n0 + u * 8;
    f32x4 bias_lo = {0, 0, 0, 0}, bias_hi = {0, 0, 0, 0};
    if (ep.bias) {
        bias_lo = *(const f32x4*)(ep.bias + n);
        bias_hi = *(const f32x4*)(ep.bias + n + 4);
    }
    const float alpha = ep.alpha;
    const int srow_l = wid * RPI + lane / LPR;                         // staging row of this lane, + it * 16
    auto pack8 = [](f32x4 x, f32x4 y) {
        bf16x8 pk = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
        return pk;
    };
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        // staging row sr of pass p is tile row (sr < 32 ? 0 : 128) + 32 p + (sr & 31)
        size_t obase[ITERS];
        [[maybe_unused]] bf16x8 pf_aux[ITERS];
        [[maybe_unused]] f32x4 pf_r0[ITERS], pf_r1[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int sr = srow_l + it * NW * RPI;
            const int m = m0 + (sr & 32) * 4 + p * 32 + (sr & 31);
            obase[it] = (size_t)m * ep.ldc + n;
            if constexpr (MODE == GEMM_DACT) pf_aux[it] = *(const bf16x8*)(ep.aux_in + obase[it]);
            if constexpr (MODE == GEMM_RESID) {
                const float* rp = ep.resid + (size_t)m * ep.ldr + n;
                pf_r0[it] = *(const f32x4*)rp;
                pf_r1[it] = *(const f32x4*)(rp + 4);
            }
        }
        __syncthreads();                       // the stage (p = 0) / the previous pass is no longer read
        {
            const int sr = wr * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int unit = (wc * 64 + j * 32 + 8 * g) / 4 + (lane >> 5);
                    const f32x4 v = {acc[p][j][4 * g], acc[p][j][4 * g + 1], acc[p][j][4 * g + 2], acc[p][j][4 * g + 3]};
                    *(f32x4*)(sC + sr * BN + (stage_slot<BN>(unit, sr) << 2)) = v;
                }
        }
        __syncthreads();
        f32x4 t0[ITERS], t1[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int rr = srow_l + it * NW * RPI;
            t0[it] = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u, rr) << 2));
            t1[it] = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u + 1, rr) << 2));
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            f32x4 v0 = t0[it] * alpha + bias_lo, v1 = t1[it] * alpha + bias_hi;
            const size_t o = obase[it];
            if constexpr (MODE == GEMM_STORE && ACT != ACT_NONE) {
                if constexpr (AUX_OUT) {
                    if (ep.aux_grad) {                            // block-uniform
                        f32x4 g0, g1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            act_and_grad_into(ACT, v0[e], v0[e], g0[e]);
                            act_and_grad_into(ACT, v1[e], v1[e], g1[e]);
                        }
                        *(bf16x8*)(ep.aux_out + o) = pack8(g0, g1);
                    } else {
                        *(bf16x8*)(ep.aux_out + o) = pack8(v0, v1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v0[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v0[e]) : gelu_erf_f(v0[e]);
                            v1[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v1[e]) : gelu_erf_f(v1[e]);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v0[e]) : gelu_erf_f(v0[e]);
                        v1[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v1[e]) : gelu_erf_f(v1[e]);
                    }
                }
            } else if constexpr (MODE == GEMM_RESID) {
                v0 += pf_r0[it];
                v1 += pf_r1[it];
            } else if constexpr (MODE == GEMM_DACT) {
                const bf16x8 pa = pf_aux[it];
                if (ep.aux_grad) {                                // block-uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] *= bf2f(pa[e]); v1[e] *= bf2f(pa[4 + e]); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = bf2f(pa[e]), x1 = bf2f(pa[4 + e]);
                        v0[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x0) : gelu_erf_grad_f(x0);
                        v1[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x1) : gelu_erf_grad_f(x1);
                    }
                }
            }
            if constexpr (OUT_F32) {
                *(f32x4*)(ep.out_f32 + o) = v0;
                *(f32x4*)(ep.out_f32 + o + 4) = v1;
            }
            if constexpr (OUT_BF16) *(bf16x8*)(ep.out_bf16 + o) = pack8(v0, v1);
        }
    }
    __syncthreads();                           // the staging rows are read: the stage may take DMA again
}

template <int MODE, int ACT, bool OUT_F32, bool OUT_BF16, bool AUX_OUT>
__global__ __launch_bounds__(512, 2) void gemm_nt2p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int M, int N,
                                                          int K, int lda, int ldb, GemmEpilogue ep, int ntiles) {
    constexpr int BKT = 64, A_BYTES = 256 * BKT * 2, STAGE = 2 * A_BYTES, TM = 128, TN = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles_n = N / 256, nk = K / BKT;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)M * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)N * (uint32_t)ldb * 2u);
    // tile sequence of this workgroup: full rounds take tile  r G + xcd_remap(b, G)  (the workgroups of an XCD share operand
    // panels in its L2), the last, partial round is spread over all XCDs: workgroups b < rem take  full G + xcd_remap(b, rem)
    const int G = gridDim.x, full = ntiles / G, rem = ntiles - full * G;
    auto tile_of = [&](int r) -> int {
        if (r < full) return r * G + xcd_remap(blockIdx.x, G);
        if (r == full && (int)blockIdx.x < rem) return full * G + xcd_remap(blockIdx.x, rem);
        return -1;
    };
    // stage chunk c (16 KB: the rows one slot reads, gemm_nt2_kernel) of the k tile at k0 of tile (m0, n0) into stage buffer sb
    auto chunk = [&](int c, int m0, int n0, int k0, char* sb, int lane) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int i = wid * 2 + t;
            int row0;
            if (c == 0 || c == 3) row0 = (i < 8 ? 0 : 128) + (c == 3 ? 64 : 0) + (i & 7) * 8;
            else row0 = (i >> 2) * 64 + (c == 2 ? 32 : 0) + (i & 3) * 8;
            const int r = row0 + (lane >> 3);
            const int cc = nt2_swz<BKT>(r, lane & 7);
            if (c == 0 || c == 3)
                glds16(rsA, sb + row0 * 128, ((uint32_t)(m0 + r) * (uint32_t)lda + (uint32_t)(k0 + cc * 8)) * 2u);
            else
                glds16(rsB, sb + A_BYTES + row0 * 128, ((uint32_t)(n0 + r) * (uint32_t)ldb + (uint32_t)(k0 + cc * 8)) * 2u);
        }
    };
    int t_cur = tile_of(0);
    int m0 = (t_cur / tiles_n) * 256, n0 = (t_cur % tiles_n) * 256;
    int par = 0;                               // LDS stage of the current k tile
    chunk(0, m0, n0, 0, smem, lane0); chunk(1, m0, n0, 0, smem, lane0); chunk(2, m0, n0, 0, smem, lane0); chunk(3, m0, n0, 0, smem, lane0);
    chunk(0, m0, n0, BKT, smem + STAGE, lane0); chunk(1, m0, n0, BKT, smem + STAGE, lane0);
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    for (int r = 0;; ++r) {
        const int t_next = tile_of(r + 1);
        const bool has_next = t_next >= 0;
        const int m0n = has_next ? (t_next / tiles_n) * 256 : 0, n0n = has_next ? (t_next % tiles_n) * 256 : 0;
        int lane = lane0;
        asm volatile("" : "+v"(lane));         // opaque per tile: keeps the lane-dependent addresses out of the tile loop's preheader
        const int arow = wr * TM + (lane & 31), brow = wc * TN + (lane & 31), cl = lane >> 5;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
        if (wr == 1) __builtin_amdgcn_s_barrier();     // stagger the lower half by one barrier
        bf16x8 a[2][4], b0[4], b1[4];
#define SPN_PSLOT_MFMA(I0, J, BREG)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        __builtin_amdgcn_s_barrier();                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        asm volatile("" : "+v"(BREG[0]), "+v"(BREG[1]), "+v"(BREG[2]), "+v"(BREG[3]));                  \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                              \
            acc[I0][J] = mfma32(BREG[kk], a[0][kk], acc[I0][J]);                                        \
            acc[I0 + 1][J] = mfma32(BREG[kk], a[1][kk], acc[I0 + 1][J]);                                \
        }                                                                                               \
        asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));                                      \
        __builtin_amdgcn_s_setprio(0);                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        __builtin_amdgcn_s_barrier();                                                                   \
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt < nk; ++kt) {
            const char* sA = smem + par * STAGE;
            const char* sB = sA + A_BYTES;
            char* s1 = smem + (par ^ 1) * STAGE;       // stage of the k tile after this one
            char* s2 = smem + par * STAGE;             // ... and of the one after that (this stage again)
            const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
            const bool v1 = in1 || has_next;                                   // a k tile follows (this tile's or the next tile's first)
            const bool v2 = in2 || (kt + 2 == nk && has_next);                 // the next tile's k tile 1 waits for the epilogue
            const int m1 = in1 ? m0 : m0n, n1 = in1 ? n0 : n0n, k1 = in1 ? (kt + 1) * BKT : 0;
            const int m2 = in2 ? m0 : m0n, n2 = in2 ? n0 : n0n, k2 = in2 ? (kt + 2) * BKT : 0;
            const bool skipw = kt == 0 && r > 0;       // everything k tile 0 of a later tile reads landed before the epilogue
            // slot 0
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b0[kk] = nt2_frag<BKT>(sB, brow, kk * 2 + cl);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a[0][kk] = nt2_frag<BKT>(sA, arow, kk * 2 + cl);
                a[1][kk] = nt2_frag<BKT>(sA, arow + 32, kk * 2 + cl);
            }
            if (v1) chunk(2, m1, n1, k1, s1, lane);
            if (!skipw) { if (v1) wait_vmcnt<8>(); else wait_vmcnt<0>(); }      // c2 of this k tile, read in slot 1
            SPN_PSLOT_MFMA(0, 0, b0)
            // slot 1
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b1[kk] = nt2_frag<BKT>(sB, brow + 32, kk * 2 + cl);
            if (v1) chunk(3, m1, n1, k1, s1, lane);
            if (!skipw) { if (v1) wait_vmcnt<8>(); else wait_vmcnt<0>(); }      // c3 of this k tile, read in slot 2
            SPN_PSLOT_MFMA(0, 1, b1)
            // slot 2
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a[0][kk] = nt2_frag<BKT>(sA, arow + 64, kk * 2 + cl);
                a[1][kk] = nt2_frag<BKT>(sA, arow + 96, kk * 2 + cl);
            }
            if (v2) chunk(0, m2, n2, k2, s2, lane);
            SPN_PSLOT_MFMA(2, 1, b1)
            // slot 3
            if (v2) chunk(1, m2, n2, k2, s2, lane);
            if (v1) { if (v2) wait_vmcnt<8>(); else wait_vmcnt<0>(); }          // c0, c1 of the following k tile, read in its slot 0
            SPN_PSLOT_MFMA(2, 0, b0)
            par ^= 1;
        }
#undef SPN_PSLOT_MFMA
        if (wr == 0) __builtin_amdgcn_s_barrier();     // re-align the two halves
        // `par` now names the stage of the next tile's k tile 0; the other one (this tile's last k tile) stages the epilogue
        nt_epilogue_p<MODE, ACT, OUT_F32, OUT_BF16, AUX_OUT>(acc, smem + (par ^ 1) * STAGE, m0, n0, wr, wc, wid, lane, ep);
        if (!has_next) break;
        m0 = m0n; n0 = n0n;
        chunk(0, m0, n0, BKT, smem + (par ^ 1) * STAGE, lane);        // the held-back first half of the new tile's k tile 1
        chunk(1, m0, n0, BKT, smem + (par ^ 1) * STAGE, lane);
    }
}

template <int MODE, int ACT, bool OUT_F32, bool OUT_BF16, bool AUX_OUT>
static int launch_nt2p(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, const GemmEpilogue& ep,
                       hipStream_t st) {
    constexpr int LDS = 131072;
    auto kern = gemm_nt2p_kernel<MODE, ACT, OUT_F32, OUT_BF16, AUX_OUT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = (M / 256) * (N / 256), cus = device_cu_count();
    hipLaunchKernelGGL(kern, dim3(tiles < cus ? tiles : cus), dim3(512), LDS, st, A, B, M, N, K, lda, ldb, ep, tiles);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// SPN_GEMM_PERSIST=0 keeps every NT product on the one-tile-per-workgroup kernel (A/B switch)
static int g_nt_persist = -1;                  // spn_gemm_config(0, v): -1 = the environment default, 0 = off, 1 = on
static bool nt_persist_on() {
    static const bool v = [] {
        const char* e = spn_env("SPN_GEMM_PERSIST");
        return e && e[0] == '1';                   // opt-in (experiments build): the step is slower with it
    }();
    return g_nt_persist < 0 ? v : g_nt_persist != 0;
}

// the persistent kernel takes multi-round products of full tiles with the epilogues the towers use; returns false otherwise
static bool dispatch_nt2p(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode, const GemmEpilogue& ep,
                          hipStream_t st, int* rc) {
    if (!nt_persist_on() || M % 256 || N % 256 || K < 128 || ep.direct_store || ep.ldc % 8) return false;
    const int tiles = (M / 256) * (N / 256);
    // Per launch (rocprofv3, profiles/r04_gemm_persist_ab.txt): 924 tiles (3.6 rounds) fc + QuickGELU 124.4 -> 118.4 us,
    // d-activation 108.7 -> 107.1; 693 tiles (2.7 rounds, plain bf16 store) 69.4 -> 76.1 us - the four-pass epilogue (eight
    // barriers instead of four) costs a plain store more than the two hidden prologues return.
    if (tiles <= device_cu_count()) return false;
    const bool b16 = ep.out_bf16 && !ep.out_f32, f32o = ep.out_f32 && !ep.out_bf16;
#define SPN_NT2P(MODE_, ACT_, F_, B_, A_) (*rc = launch_nt2p<MODE_, ACT_, F_, B_, A_>(A, B, M, N, K, lda, ldb, ep, st), true)
    if (mode == GEMM_STORE && ep.act == ACT_NONE && b16) return SPN_NT2P(GEMM_STORE, ACT_NONE, false, true, false);
    if (mode == GEMM_STORE && ep.act == ACT_NONE && f32o) return SPN_NT2P(GEMM_STORE, ACT_NONE, true, false, false);
    if (mode == GEMM_STORE && ep.act == ACT_QUICKGELU && b16 && ep.aux_out) return SPN_NT2P(GEMM_STORE, ACT_QUICKGELU, false, true, true);
    if (mode == GEMM_STORE && ep.act == ACT_GELU_ERF && b16 && ep.aux_out) return SPN_NT2P(GEMM_STORE, ACT_GELU_ERF, false, true, true);
    if (mode == GEMM_DACT && ep.act == ACT_QUICKGELU && b16 && ep.aux_in) return SPN_NT2P(GEMM_DACT, ACT_QUICKGELU, false, true, false);
    if (mode == GEMM_DACT && ep.act == ACT_GELU_ERF && b16 && ep.aux_in) return SPN_NT2P(GEMM_DACT, ACT_GELU_ERF, false, true, false);
    if (mode == GEMM_RESID && f32o && ep.resid && ep.ldr % 4 == 0) return SPN_NT2P(GEMM_RESID, ACT_NONE, true, false, false);
#undef SPN_NT2P
    return false;
}

#endif  // SPN_EXPERIMENTS (persistent NT kernel)

#ifdef SPN_EXPERIMENTS   // the hand-scheduled 4-wave kernel: measured slower in the step (LABNOTES.md 5.3), not in the shipped library
// Epilogue of gemm_nt3: FULL 256x256 tiles only (the launcher routes anything else to gemm_nt2), straight-line code, and
// wave-private: wave (wr, wc) owns the 128x128 sub-tile rows wr*128.., columns wc*128.. and turns it row-major through
// its OWN 32 KB of LDS, so after the one barrier that ends the k loop no wave waits for another - with one wave per SIMD
// every exposed latency is idle time (the shared, branchy epilogue of gemm_nt2 took 15 k cycles per tile here, as long
// as the k loop of a K = 768 product; 30 k with the activation copy).
//   bf16 results (GEMM_STORE): bias / activation in the MFMA layout, packed to bf16, ONE pass over 128 rows x 256 B:
//     8-byte ds_write of a lane's 4 columns, 16-byte ds_read + 16-byte global store of whole 256-B row segments.
//   fp32 staging (GEMM_RESID, GEMM_DACT: the row-major operand - residual stream / pre-activation - is loaded coalesced
//     and applied after the transposition): two passes of 64 rows x 512 B.
// 16-byte units are XOR-swizzled by the row, conflict-free for the column-wise writes and the row-wise reads.
template <int MODE, int ACT>
__device__ __forceinline__ void nt3_epilogue(f32x16 (&acc)[4][4], char* smem, int m0, int n0, int wr, int wc, int wid,
                                             int lane, const GemmEpilogue& ep) {
    char* sW = smem + wid * 32768;
    const int l31 = lane & 31, half = lane >> 5;
    const int mw = m0 + wr * 128, nw = n0 + wc * 128;          // first row / column of this wave's sub-tile
    __syncthreads();                                           // every wave is done with the operand tiles
    if constexpr (MODE == GEMM_STORE) {
        // bias of this lane's columns nw + j*32 + 8g + 4*half + 0..3
        f32x4 bias[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bias[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ep.bias) bias[j][g] = *(const f32x4*)(ep.bias + nw + j * 32 + 8 * g + 4 * half);
            }
        constexpr int PASSES = ACT != ACT_NONE ? 2 : 1;        // pass 0 of an activation GEMM: the pre-activation copy
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
            bf16_t* dst = (ACT != ACT_NONE && pass == 0) ? ep.aux_out : ep.out_bf16;
            const bool want_grad = ACT != ACT_NONE && pass == 0 && ep.aux_grad;      // wave-uniform
            if (dst == nullptr) continue;                      // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 32 + l31;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        v = v * ep.alpha + bias[j][g];
                        if (ACT != ACT_NONE && (pass == 1 || want_grad)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float y, gr;
                                act_and_grad_into(ACT, v[e], y, gr);
                                v[e] = pass == 1 ? y : gr;
                            }
                        }
                        const bf16x4 pk = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                        *(bf16x4*)(sW + r * 256 + (((j * 4 + g) ^ (r & 15)) << 4) + half * 8) = pk;
                    }
            }
            // row-major: 16 lanes x 16 B = one 256-B row segment, 4 rows per instruction
            const int u = lane & 15, rq = lane >> 4;
            bf16x8 t[32];
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int r = it * 4 + rq;
                t[it] = *(const bf16x8*)(sW + r * 256 + ((u ^ (r & 15)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int r = it * 4 + rq;
                *(bf16x8*)(dst + (size_t)(mw + r) * ep.ldc + nw + u * 8) = t[it];
            }
        }
    } else {
        // fp32 staging, 2 passes of 64 rows: lane l of the read-back owns columns nw + 4*(l&31) .. +3 of rows 2*it + (l>>5)
        const int u = lane & 31, rh = lane >> 5;
        const int n = nw + u * 4;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (ep.bias) bias = *(const f32x4*)(ep.bias + n);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = c * 2 + ii;
                const int r = ii * 32 + l31;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *(f32x4*)(sW + r * 512 + (((j * 8 + 2 * g + half) ^ (r & 31)) << 4)) = v;
                    }
            }
            // batches of 8 row pairs: the row-major operand of the next batch is in flight while this one is finished
            constexpr int NB = 4, PER = 8;
            [[maybe_unused]] f32x4 rs[2][PER];
            [[maybe_unused]] bf16x4 ax[2][PER];
            auto fetch = [&](int b, int slot) {
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int r = (b * PER + q) * 2 + rh;
                    const size_t m = (size_t)(mw + c * 64 + r);
                    if constexpr (MODE == GEMM_RESID) rs[slot][q] = *(const f32x4*)(ep.resid + m * ep.ldr + n);
                    if constexpr (MODE == GEMM_DACT) ax[slot][q] = *(const bf16x4*)(ep.aux_in + m * ep.ldc + n);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) fetch(b + 1, (b + 1) & 1);
                f32x4 v[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int r = (b * PER + q) * 2 + rh;
                    v[q] = *(const f32x4*)(sW + r * 512 + ((u ^ (r & 31)) << 4));
                }
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int r = (b * PER + q) * 2 + rh;
                    const size_t m = (size_t)(mw + c * 64 + r);
                    f32x4 x = v[q] * ep.alpha + bias;
                    if constexpr (MODE == GEMM_RESID) x += rs[b & 1][q];
                    if constexpr (MODE == GEMM_DACT) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float p = bf2f(ax[b & 1][q][e]);
                            x[e] *= ep.aux_grad ? p : (ACT == ACT_QUICKGELU ? quick_gelu_grad_f(p) : gelu_erf_grad_f(p));
                        }
                    }
                    if (ep.out_f32) *(f32x4*)(ep.out_f32 + m * ep.ldc + n) = x;
                    if (ep.out_bf16) {
                        const bf16x4 pk = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3])};
                        *(bf16x4*)(ep.out_bf16 + m * ep.ldc + n) = pk;
                    }
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------- NT, third generation
// gemm_nt3: 256x256x64 tile, FOUR waves (2 x 2, one per SIMD, wave tile 128x128 = all 256 AGPRs), hand-scheduled main
// loop (tools/gen_gemm3.py -> gemm3_nt_loop.inc, one inline-asm statement: DMA, fragment reads and MFMA interleaved by
// hand, one barrier per k tile).  Same LDS image (nt2_swz) and the same epilogue as gemm_nt2.
template <int MODE, int ACT, int VAR>
__global__ __launch_bounds__(256, 1) void gemm_nt3_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                          int M, int N, int K, int lda, int ldb, GemmEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint64_t dbg_t0 = (SPN_DBG(ep) & 64) ? __builtin_readcyclecounter() : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = (N + 255) / 256;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * 256, n0 = (bid % tiles_n) * 256;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)M * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)N * (uint32_t)ldb * 2u);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = zero16();
    const int nk = K / 64;
    const uint32_t lds_base = (uint32_t)(uintptr_t)LDS_PTR(smem);
    uint32_t cyc, ticks;
    const uint64_t dbg_t1 = (SPN_DBG(ep) & 64) ? __builtin_readcyclecounter() : 0;
#define SPN_NT3_OPERANDS                                                                                              \
        : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),       \
          "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]),       \
          "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]), "=s"(cyc), "=s"(ticks)                  \
        : "v"(tid), "s"(rsA), "s"(rsB), "s"(m0), "s"(n0), "s"(lda), "s"(ldb), "s"(nk), "s"(lds_base)                  \
        : SPN_GEMM3_CLOBBERS
    if constexpr (VAR == 0) {
        asm volatile(
#include "gemm3_nt_loop.inc"
            SPN_NT3_OPERANDS);
    }
#ifdef SPN_NT3_ABL   // experiment builds only (tools/gen_gemm3.py --ablation, tools/build_variant.sh): results are wrong by design
    else if constexpr (VAR == 1) {
        asm volatile(
#include "gemm3_nt_loop_nodma.inc"
            SPN_NT3_OPERANDS);
    } else if constexpr (VAR == 2) {
        asm volatile(
#include "gemm3_nt_loop_noread.inc"
            SPN_NT3_OPERANDS);
    } else if constexpr (VAR == 3) {
        asm volatile(
#include "gemm3_nt_loop_nobar.inc"
            SPN_NT3_OPERANDS);
    } else {
        asm volatile(
#include "gemm3_nt_loop_mfma.inc"
            SPN_NT3_OPERANDS);
    }
#endif
#undef SPN_NT3_OPERANDS
    // SPN_GEMM_DBG bit 32: clock probe of the k loop of the last tile (same convention as gemm_nt2_kernel)
    if ((SPN_DBG(ep) & 32) && blockIdx.x == gridDim.x - 1 && tid == 0 && ep.out_bf16) {
        uint32_t* o = (uint32_t*)ep.out_bf16;
        o[0] = cyc;
        o[1] = ticks;
        return;
    }
    const uint64_t dbg_t2 = (SPN_DBG(ep) & 64) ? __builtin_readcyclecounter() : 0;
    nt3_epilogue<MODE, ACT>(acc, smem, m0, n0, wr, wc, wid, lane, ep);
    // SPN_GEMM_DBG bit 64: phase times of the first block (shader cycles): setup, asm statement (prologue DMA + k loop),
    // epilogue, and the k loop alone - written over the first 16 bytes of the fp32 / bf16 output
    if ((SPN_DBG(ep) & 64) && blockIdx.x == 0 && tid == 0) {
        const uint64_t t3 = __builtin_readcyclecounter();
        uint32_t* o = ep.out_f32 ? (uint32_t*)ep.out_f32 : (uint32_t*)ep.out_bf16;
        o[0] = (uint32_t)(dbg_t1 - dbg_t0);
        o[1] = (uint32_t)(dbg_t2 - dbg_t1);
        o[2] = (uint32_t)(t3 - dbg_t2);
        o[3] = cyc;
    }
}

template <int MODE, int ACT, int VAR = 0>
static int launch_nt3(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, const GemmEpilogue& ep,
                      hipStream_t st) {
    constexpr int LDS = 131072;
    auto kern = gemm_nt3_kernel<MODE, ACT, VAR>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), LDS, st, A, B, M, N, K, lda, ldb, ep);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// gemm_nt3 handles full 256x256 tiles with 16-byte aligned rows; everything else stays on gemm_nt2
static bool nt3_ok(int M, int N, int mode, const GemmEpilogue& ep) {
    if (M % 256 || N % 256 || ep.ldc % 8 || (ep.resid && ep.ldr % 4)) return false;
    if (mode == GEMM_STORE) return ep.out_f32 == nullptr && ep.out_bf16 != nullptr;      // bf16 staging
    return mode == GEMM_RESID || mode == GEMM_DACT;
}

static int dispatch_nt3(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode,
                        const GemmEpilogue& ep, hipStream_t st) {
#define SPN_NT3(MODE_, ACT_) launch_nt3<MODE_, ACT_>(A, B, M, N, K, lda, ldb, ep, st)
#ifdef SPN_NT3_ABL
    // SPN_NT3_VAR (-DSPN_NT3_ABL experiment builds, plain-store kernel only): ablation variants of the main loop, results
    // are wrong by design - 1 no DMA, 2 no fragment reads, 3 no barrier, 4 MFMA only.  Not in the shipped library.
    static const int var = [] {
        const char* e = spn_env("SPN_NT3_VAR");
        return e ? atoi(e) : 0;
    }();
    if (mode == GEMM_STORE && ep.act == ACT_NONE && var) {
        switch (var) {
            case 1: return launch_nt3<GEMM_STORE, ACT_NONE, 1>(A, B, M, N, K, lda, ldb, ep, st);
            case 2: return launch_nt3<GEMM_STORE, ACT_NONE, 2>(A, B, M, N, K, lda, ldb, ep, st);
            case 3: return launch_nt3<GEMM_STORE, ACT_NONE, 3>(A, B, M, N, K, lda, ldb, ep, st);
            default: return launch_nt3<GEMM_STORE, ACT_NONE, 4>(A, B, M, N, K, lda, ldb, ep, st);
        }
    }
#endif
    if (mode == GEMM_STORE) {
        if (ep.act == ACT_NONE) return SPN_NT3(GEMM_STORE, ACT_NONE);
        if (ep.act == ACT_QUICKGELU) return SPN_NT3(GEMM_STORE, ACT_QUICKGELU);
        if (ep.act == ACT_GELU_ERF) return SPN_NT3(GEMM_STORE, ACT_GELU_ERF);
        return SPN_ERR_ARG;
    }
    if (mode == GEMM_RESID) {
        if (!ep.resid || !ep.out_f32) return SPN_ERR_ARG;
        return SPN_NT3(GEMM_RESID, ACT_NONE);
    }
    if (mode == GEMM_DACT) {
        if (!ep.aux_in) return SPN_ERR_ARG;
        if (ep.act == ACT_QUICKGELU) return SPN_NT3(GEMM_DACT, ACT_QUICKGELU);
        if (ep.act == ACT_GELU_ERF) return SPN_NT3(GEMM_DACT, ACT_GELU_ERF);
    }
#undef SPN_NT3
    return SPN_ERR_ARG;
}

#endif  // SPN_EXPERIMENTS

#ifdef SPN_EXPERIMENTS
int gemm_nt_persist_set(int v) { g_nt_persist = v < 0 ? -1 : (v ? 1 : 0); return SPN_OK; }
#else
int gemm_nt_persist_set(int v) { return v > 0 ? SPN_ERR_ARG : SPN_OK; }     // the kernel is not part of this build
#endif

static bool nt_phased() {
    static const bool v = [] {
        const char* e = spn_env("SPN_GEMM_NT_PHASED");
        return !(e && e[0] == '0');
    }();
    return v;
}

static bool tn_phased() {
    static const bool v = [] {
        const char* e = spn_env("SPN_GEMM_TN_PHASED");
        return !(e && e[0] == '0');
    }();
    return v;
}

static bool gemm_spread() {
    static const bool v = [] {
        const char* e = spn_env("SPN_GEMM_SPREAD");
        return !(e && e[0] == '0');
    }();
    return v;
}

template <int BM, int BN, int WM, int WN, int STAGES, int BKT = 64, bool PHASED = false>
static int dispatch_nt2(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode,
                        const GemmEpilogue& ep, hipStream_t st) {
#define SPN_NT2(MODE_, ACT_)                                                                                   \
    (PHASED ? launch_nt2<BM, BN, WM, WN, STAGES, MODE_, ACT_, PHASED ? 2 : 1, BKT>(A, B, M, N, K, lda, ldb, ep, st) \
            : gemm_spread() ? launch_nt2<BM, BN, WM, WN, STAGES, MODE_, ACT_, 1, BKT>(A, B, M, N, K, lda, ldb, ep, st) \
                            : launch_nt2<BM, BN, WM, WN, STAGES, MODE_, ACT_, 0, BKT>(A, B, M, N, K, lda, ldb, ep, st))
    if (mode == GEMM_STORE) {
        if (ep.act == ACT_NONE) return SPN_NT2(GEMM_STORE, ACT_NONE);
        if (ep.act == ACT_QUICKGELU) return SPN_NT2(GEMM_STORE, ACT_QUICKGELU);
        if (ep.act == ACT_GELU_ERF) return SPN_NT2(GEMM_STORE, ACT_GELU_ERF);
        return SPN_ERR_ARG;
    }
    if (mode == GEMM_RESID) {
        if (!ep.resid || !ep.out_f32) return SPN_ERR_ARG;
        return SPN_NT2(GEMM_RESID, ACT_NONE);
    }
    if (mode == GEMM_DACT) {
        if (!ep.aux_in) return SPN_ERR_ARG;
        if (ep.act == ACT_QUICKGELU) return SPN_NT2(GEMM_DACT, ACT_QUICKGELU);
        if (ep.act == ACT_GELU_ERF) return SPN_NT2(GEMM_DACT, ACT_GELU_ERF);
    }
#undef SPN_NT2
    return SPN_ERR_ARG;
}

// Bank-row width of a statistics tile: 256, or 128 for narrow banks (D <= 256: the BLIP / BLIP-2 heads).  Measured (round 4,
// tools/bank_bench.py, kernel-only): B = 128, M = 30 000, D = 256: 21.2 -> 14.5 us with 128-row tiles (a 256-row tile of a
// 256-wide bank is 118 tiles of little work each; 235 fill the chip) - but B = 256, M = 40 000, D = 768: 43.7 -> 48.9 us (the
// 3-stage one-barrier loop of the 256 x 128 configuration loses more than the finer tiling gains).  SPN_BANK_GEMM_BN=128 / 256
// forces one.  The tile count is what sizes the partial buffers: callers ask gemm_bank_stats_tiles(M, D).
int gemm_bank_stats_bn(int D) {
    static const int forced = [] {
        const char* e = spn_env("SPN_BANK_GEMM_BN");
        return !e ? 0 : (e[0] == '2' ? 256 : (e[0] == '1' ? 128 : 0));
    }();
    return forced ? forced : (D <= 256 ? 128 : 256);
}
int gemm_bank_stats_tiles(int M, int D) { const int bn = gemm_bank_stats_bn(D); return (M + bn - 1) / bn; }

int gemm_bank_stats(const bf16_t* q, const bf16_t* bank, int B, int M, int D, int ldq, int ldb, const int64_t* labels,
                    float inv_tau, int m_begin, float* partial, hipStream_t st, bf16_t* p_out, int ldp, float* max_out) {
    if (B <= 0 || M <= 0 || D <= 0 || !labels || !partial) return SPN_ERR_ARG;
    if (D % BK2 || ldq % 8 || ldb % 8) return SPN_ERR_SHAPE;
    if ((uint64_t)B * ldq * 2 >= (1ull << 32) || (uint64_t)M * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    GemmEpilogue e;
    e.bs_labels = labels; e.bs_out = partial; e.bs_inv_tau = inv_tau; e.bs_m_begin = m_begin;
    if (p_out && (ldp % 256 || ldp < M || !max_out)) return SPN_ERR_ARG;
    e.bs_p_out = p_out; e.bs_ldp = ldp; e.bs_max_out = max_out;
    e.ldc = 8;
    if (gemm_bank_stats_bn(D) == 128)
        return launch_nt2<256, 128, 4, 2, 3, GEMM_BANKSTATS, ACT_NONE, 1, 64>(q, bank, B, M, D, ldq, ldb, e, st);
    return launch_nt2<256, 256, 2, 4, 2, GEMM_BANKSTATS, ACT_NONE, 2, 64>(q, bank, B, M, D, ldq, ldb, e, st);
}

int gemm_nt2(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode, const GemmEpilogue& ep,
             hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return SPN_ERR_ARG;
    if (K % BK2 || N % 4 || lda % 8 || ldb % 8 || ep.ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)M * lda * 2 >= (1ull << 32) || (uint64_t)N * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (!ep.out_f32 && !ep.out_bf16) return SPN_ERR_ARG;
    static const int direct = [] {
        const char* e = spn_env("SPN_GEMM_EPI_DIRECT");
        return (e && e[0] == '1') ? 1 : 0;
    }();
    // SPN_GEMM_DBG: in-kernel probes / bottleneck-elimination switches (some overwrite the first bytes of the output):
    // compiled only into -DSPN_GEMM_PROBES experiment builds (tools/build_variant.sh), never into the shipped library
#ifdef SPN_GEMM_PROBES
    static const int dbg = [] {
        const char* e = spn_env("SPN_GEMM_DBG");
        return e ? atoi(e) : 0;
    }();
#else
    constexpr int dbg = 0;
#endif
    GemmEpilogue e2 = ep;
    e2.direct_store = direct;
    e2.dbg = dbg;
    ProfScope prof(PK_GEMM_NT, 2.0 * M * N * K, st);
    switch (gemm_cfg()) {
#ifdef SPN_EXPERIMENTS
        case 7:                                                                   // 4 waves, hand-scheduled loop
            if (nt3_ok(M, N, mode, e2)) return dispatch_nt3(A, B, M, N, K, lda, ldb, mode, e2, st);
            return dispatch_nt2<256, 256, 2, 4, 2, 64, true>(A, B, M, N, K, lda, ldb, mode, e2, st);
#endif
        case 1: return dispatch_nt2<256, 128, 4, 2, 3>(A, B, M, N, K, lda, ldb, mode, e2, st);
        case 3:   // default: 256x256x64, 8 waves; SPN_GEMM_NT_PHASED=0 selects the one-barrier-per-k-tile loop
#ifdef SPN_EXPERIMENTS
            {
                // SPN_GEMM_TWO_WG=1 (experiments build): the short-k, wide products (qkv / fc / d-activation: K <= 768, N >= 2048) on
                // 128 x 256 x 32 tiles with three 24 KB stages - 72 KB of LDS and <= 128 registers per wave, so that TWO workgroups
                // share a CU and one's epilogue runs under the other's k loop.  Measured in the step (rocprofv3, round 4): fc +
                // QuickGELU 119.6 -> 147.2 us, d-activation 108.6 -> 131.4, qkv 70 -> 87.8, step 13.28 -> 14.09 ms: the one-barrier
                // loop on a tile with 1.5x the LDS-DMA bytes per flop loses more than the hidden epilogue returns.
                static const int two_wg = [] { const char* v = spn_env("SPN_GEMM_TWO_WG"); return v ? atoi(v) : 0; }();
                if (two_wg && K <= 768 && N >= 2048 && M % 128 == 0 && N % 256 == 0 && K % 32 == 0)
                    return dispatch_nt2<128, 256, 2, 4, 3, 32>(A, B, M, N, K, lda, ldb, mode, e2, st);
            }
#endif
            if (nt_phased()) {
#ifdef SPN_EXPERIMENTS
                int rc = SPN_OK;
                if (dispatch_nt2p(A, B, M, N, K, lda, ldb, mode, e2, st, &rc)) return rc;      // multi-round: persistent walk (opt-in)
#endif
                return dispatch_nt2<256, 256, 2, 4, 2, 64, true>(A, B, M, N, K, lda, ldb, mode, e2, st);
            }
            return dispatch_nt2<256, 256, 2, 4, 2>(A, B, M, N, K, lda, ldb, mode, e2, st);
        case 6: return dispatch_nt2<256, 256, 2, 4, 2, 64, true>(A, B, M, N, K, lda, ldb, mode, e2, st);   // 8-slot staggered
        default: return dispatch_nt2<256, 256, 4, 2, 2>(A, B, M, N, K, lda, ldb, mode, e2, st);
    }
}

// Mid-size NT tiles on the second-generation kernel (v_mfma_f32_32x32x16_bf16, LDS-DMA ring), for products with too few
// 256 x 256 tiles to fill the chip (the packed text tower's ~5 k rows, 32 triplets per GPU under 8-way data parallelism: 2 464
// rows, the BERT side of the BLIP step: 4 096 rows) - gemm_nt() routes them here when SPN_NT_MID selects a variant:
//   1: 128 x 128 x 64, 4 waves (2 x 2, wave tile 64 x 64), 2 stages = 64 KB  -> two workgroups per CU
//   2: 128 x 128 x 64, 4 waves, 3 stages = 96 KB                              -> one workgroup per CU, counted vmcnt
//   3: 128 x 256 x 64, 8 waves (2 x 4, wave tile 64 x 64), 2 stages = 96 KB
//   4: 64 x 256 x 64, 4 waves (1 x 4), 2 stages = 80 KB                       -> two workgroups per CU
// Measured and NOT kept (round 5, tools/small_gemm_bench.py, profiles/r05_small_gemm_mid_tiles.txt; sum of the qkv / out / fc + GELU /
// c_proj products at 2 464 / 4 096 / 5 248 rows): the routed 128 x 128 kernel of gemm.hip 82.3 / 97.5 / 103.3 us; variant 1: 97.1 /
// 106.3 / 112.0; 2: 91.3 / 96.1 / 104.3; 3: 107.8 / 115.2 / 121.3; 4: 101.1 / 109.1 / 114.4 - at 10-30 us per launch these products
// are bound by fill / drain of a single round, not by the tile's steady state.  Experiments build only.
#ifdef SPN_EXPERIMENTS
int gemm_nt2_mid(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode, const GemmEpilogue& ep,
                 hipStream_t st, int variant) {
    if (M <= 0 || N <= 0 || K <= 0) return SPN_ERR_ARG;
    if (K % BK2 || N % 4 || lda % 8 || ldb % 8 || ep.ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)M * lda * 2 >= (1ull << 32) || (uint64_t)N * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (!ep.out_f32 && !ep.out_bf16) return SPN_ERR_ARG;
    GemmEpilogue e2 = ep;
    e2.direct_store = 0;
    e2.dbg = 0;
    ProfScope prof(PK_GEMM_NT, 2.0 * M * N * K, st);
    switch (variant) {
        case 1: return dispatch_nt2<128, 128, 2, 2, 2>(A, B, M, N, K, lda, ldb, mode, e2, st);
        case 2: return dispatch_nt2<128, 128, 2, 2, 3>(A, B, M, N, K, lda, ldb, mode, e2, st);
        case 3: return dispatch_nt2<128, 256, 2, 4, 2>(A, B, M, N, K, lda, ldb, mode, e2, st);
        case 4: return dispatch_nt2<64, 256, 1, 4, 2>(A, B, M, N, K, lda, ldb, mode, e2, st);
        default: return SPN_ERR_ARG;
    }
}
#else
int gemm_nt2_mid(const bf16_t*, const bf16_t*, int, int, int, int, int, int, const GemmEpilogue&, hipStream_t, int) { return SPN_ERR_ARG; }
#endif

// ----------------------------------------------------------------------------------- TN
// LDS image of a [64 k][COLS] bf16 tile: row k at byte k*2*COLS; the 32-byte chunk holding logical
// columns 16c..16c+15 sits at chunk position c ^ ((k&3)<<1): the two 16-lane groups of a half-wave
// read chunks c, c+1 of rows k0..k0+3 -> 8 distinct 32-byte bank groups.
template <int COLS, int PER_WAVE>
__device__ __forceinline__ void tn2_stage(__amdgpu_buffer_rsrc_t rs, char* sT, int kbase, int ld, int col0, int wid,
                                          int lane) {
    constexpr int ROWB = COLS * 2;
    constexpr int ROWS_PER_INSTR = 1024 / ROWB;        // 2 (256 cols) or 4 (128 cols)
    constexpr int LANES_PER_ROW = 64 / ROWS_PER_INSTR;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int R0 = (wid * PER_WAVE + i) * ROWS_PER_INSTR;
        const int r = R0 + lane / LANES_PER_ROW;
        const int pos16 = lane % LANES_PER_ROW;
        const int c32 = (pos16 >> 1) ^ ((r & 3) << 1);
        glds16(rs, sT + R0 * ROWB,
               ((uint32_t)(kbase + r) * (uint32_t)ld + (uint32_t)(col0 + c32 * 16 + (pos16 & 1) * 8)) * 2u);
    }
}

// fragment for columns [cb, cb+32) and k rows [kk*16, kk*16+16): lane l gets column cb + (l&31),
// k = kk*16 + (l>>5)*8 + 0..7
template <int COLS>
__device__ __forceinline__ bf16x8 tn2_frag(const char* sT, int cb, int kk, int lane) {
    constexpr int ROWB = COLS * 2;
    union { s16x4 h[2]; bf16x8 v; } u;
    const int g = lane >> 4, i16 = lane & 15;
    const int col = cb + (g & 1) * 16 + (i16 & 3) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int krow = kk * 16 + (g >> 1) * 8 + h * 4 + (i16 >> 2);
        const int p32 = (col >> 4) ^ ((krow & 3) << 1);
        u.h[h] = lds_tr16_b64(sT + krow * ROWB + p32 * 32 + (col & 15) * 2);
    }
    return u.v;
}

// Optional second problem of the same launch (same Kr, same split): its tiles follow the first problem's in the grid.
// A weight gradient with a small output (768 x 768: 9 tiles) cannot fill the chip on its own even with split-K, but it
// rides along with a larger one at no extra launch: tiles1 = INT_MAX when there is none.
struct TnSecond {
    const bf16_t* A; const bf16_t* B;
    int N1, N2, lda, ldb;
    float* C; int ldc;
    size_t split_stride;
    float* colsum_out;
    int tiles1;
};

// One output tile (m0, n0) of C = A^T B over the reduction rows [kb, ke): Cz / colsum_out already point at the slab this
// workgroup owns (the final matrix, or its split-K partial).  Shared by the per-problem, paired and grouped launches.
template <int BM, int BN, int WM, int WN, int STAGES, int SCHED>
__device__ __forceinline__ void tn2_tile(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int Kr, int N1, int N2,
                                         int lda, int ldb, float* __restrict__ Cz, int ldc, float* __restrict__ colsum_out,
                                         int m0, int n0, int kb, int ke, uint32_t* pace = nullptr,
                                         uint32_t pace_members = 0, int cs_mod = 0, int cs_rem = 0, int pace_lag = 0,
                                         float* __restrict__ cs_part = nullptr) {
    constexpr int NW = WM * WN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;   // 64 k rows x cols x 2 B
    constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NJ = TN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)Kr * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)Kr * (uint32_t)ldb * 2u);
    // column sums of A (the bias gradient).  cs_mod == 0: the n0 == 0 tile of a tile row computes them over all of k (its
    // wc == 0 waves: four extra MFMA per slot, +25 % matrix work for that tile).  cs_mod > 0 (grouped launch, tile rows whose
    // tiles are all whole): the cs_mod tiles of the row share the work - tile cs_rem takes the k tiles with kt % cs_mod ==
    // cs_rem, wave (wr, wc) the 32 columns i == wc - and add their partial sums to colsum_out (zeroed before the launch)
    // atomically: every tile of the launch then carries nearly the same matrix work, which k pacing needs.
    const bool cs_spread = colsum_out != nullptr && cs_mod > 0;
    const bool do_colsum = colsum_out != nullptr && cs_mod == 0 && n0 == 0 && wc == 0;
    static_assert(SCHED != 2 || (MI == 4 && WN == 4), "spread column sums: one 32-column block per wave column");

    f32x16 acc[MI][NJ], accs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        accs[i] = zero16();
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = zero16();
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

// SPN_TN_EXP (experiment builds, tools/build_variant.sh; results are wrong): 1 = no epilogue (k loop only), 2 = no k loop
#ifndef SPN_TN_EXP
#define SPN_TN_EXP 0
#endif
    const int nk = SPN_TN_EXP == 2 ? 0 : (ke - kb + BK2 - 1) / BK2;
    auto stage = [&](int kt, int buf) {
        char* s = smem + buf * STAGE;
        tn2_stage<BM, GA>(rsA, s, kb + kt * BK2, lda, m0, wid, lane);
        tn2_stage<BN, GB>(rsB, s + A_BYTES, kb + kt * BK2, ldb, n0, wid, lane);
    };
    if constexpr (SCHED == 2) {
        // Staggered schedule as in gemm_nt2_kernel (SCHED 2), with the k16 step as the slot: the LDS image is
        // k-major, so chunk c of a k tile = its k rows 16c..16c+15 of A and of B (16 KB, 2 DMA per wave) and slot c
        // reads exactly that chunk (4 + 2 transpose-read fragments) for 8 MFMA over all of the wave's 128x64
        // output.  Chunk c of k tile j+2 is staged in slot c+2 of k tile j (>= 2 slots after its last read);
        // five chunks = 10 DMA per wave stay in flight behind each wait: vmcnt(10).
        static_assert(BM == 256 && BN == 256 && WM == 2 && WN == 4 && STAGES == 2, "phased schedule geometry");
        auto chunk = [&](int c, int j) {
            char* sb = smem + (j & 1) * STAGE;
            const int R0 = c * 16 + wid * 2;
            const int r = R0 + (lane >> 5), pos16 = lane & 31;
            const int c32 = (pos16 >> 1) ^ ((r & 3) << 1);
            const uint32_t kr = (uint32_t)(kb + j * BK2 + r);
            const uint32_t co = (uint32_t)(c32 * 16 + (pos16 & 1) * 8);
            glds16(rsA, sb + R0 * 512, (kr * (uint32_t)lda + (uint32_t)m0 + co) * 2u);
            glds16(rsB, sb + A_BYTES + R0 * 512, (kr * (uint32_t)ldb + (uint32_t)n0 + co) * 2u);
        };
        auto wait_chunks = [&](bool full) {
            if (full) wait_vmcnt<10>();
            else wait_vmcnt<0>();
        };
        if (nk > 0) { chunk(0, 0); chunk(1, 0); chunk(2, 0); chunk(3, 0); }
        if (nk > 1) { chunk(0, 1); chunk(1, 1); }
        wait_chunks(nk > 1);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        // k pacing (grouped launch): the workgroups of one XCD round share A / B panels through that XCD's 4 MB L2, which
        // only works while they read the same k range at about the same time - over 308 k tiles they drift apart and
        // every tile ends up streaming its own panels from HBM (2.4x the algorithmic bytes, PMC FETCH_SIZE).  Every
        // PACE_R k tiles wave 0 adds one to the group's counter (fire and forget) and loads it; ONE k tile later it looks
        // at the loaded value (long since back: the check costs no stall) and holds the workgroup - the other waves
        // stop at the next slot barrier - until every member has reached the previous phase: the spread stays below
        // 2 PACE_R k tiles.  Correctness never depends on it (bounded spin; dispatch order makes every member resident
        // before any later block needs a CU).
        constexpr int PACE_R = 2;
        uint32_t pace_val = 0;
        int cs_left = cs_rem;                           // k tiles until this tile's next column-sum turn
        for (int kt = 0; kt < nk; ++kt, cs_left = cs_left == 0 ? cs_mod - 1 : cs_left - 1) {
            if (pace != nullptr && wid == 0) {
                if ((kt & (PACE_R - 1)) == 0) {
                    if (lane == 0) {
                        __hip_atomic_fetch_add(pace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pace_val = __hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else if ((kt & (PACE_R - 1)) == 1 && kt > PACE_R * (1 + pace_lag)) {
                    const uint32_t target = pace_members * (uint32_t)(kt / PACE_R - pace_lag);
                    uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)pace_val);
                    for (int tries = 0; v < target && tries < 20000; ++tries) {
                        __builtin_amdgcn_s_sleep(8);
                        uint32_t x = 0;
                        if (lane == 0) x = __hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
                    }
                }
            }
            const char* sA = smem + (kt & 1) * STAGE;
            const char* sB = sA + A_BYTES;
            const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = tn2_frag<BM>(sA, wr * TM + i * 32, kk, lane);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = tn2_frag<BN>(sB, wc * TN + j * 32, kk, lane);
                // slot kk of k tile kt issues chunk (kk+2)&3 of k tile kt+1 (kk < 2) or kt+2 (kk >= 2)
                if (kk < 2 ? n1 : n2) chunk((kk + 2) & 3, kk < 2 ? kt + 1 : kt + 2);
                // ... and retires the chunk of the NEXT slot: 5 newer chunks are in flight behind it
                if (kk < 3 || n1) wait_chunks(kk < 2 ? n1 : n2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(b[0]), "+v"(b[1]));
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(b[j], a[i], acc[i][j]);
                if (do_colsum) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) accs[i] = mfma32(ones, a[i], accs[i]);
                } else if (cs_spread && cs_left == 0) {
                    if (wc == 0) accs[0] = mfma32(ones, a[0], accs[0]);
                    else if (wc == 1) accs[0] = mfma32(ones, a[1], accs[0]);
                    else if (wc == 2) accs[0] = mfma32(ones, a[2], accs[0]);
                    else accs[0] = mfma32(ones, a[3], accs[0]);
                }
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[0][1]),
                             "+v"(acc[1][1]), "+v"(acc[2][1]), "+v"(acc[3][1]));
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) stage(s, s);
    int cur = 0, fill = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (STAGES >= 3 && kt + 1 < nk) wait_vmcnt<(STAGES - 2) * (GA + GB)>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, fill);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = tn2_frag<BM>(sA, wr * TM + i * 32, kk, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = tn2_frag<BN>(sB, wc * TN + j * 32, kk, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(b[j], a[i], acc[i][j]);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < MI; ++i) accs[i] = mfma32(ones, a[i], accs[i]);
            }
        }
        cur = cur == STAGES - 1 ? 0 : cur + 1;
        fill = fill == STAGES - 1 ? 0 : fill + 1;
    }
    }
    if (SPN_TN_EXP == 1) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 12345.678f) Cz[0] = t;
        return;
    }
    if (do_colsum && lane < 32) {
        // every row of the ones-product equals the column sum; lanes 0..31 hold columns m
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wr * TM + i * 32 + lane;
            if (m < N1) colsum_out[m] = accs[i][0];
        }
    }
    if (cs_spread && lane < 32) {
        const int m = m0 + wr * TM + wc * 32 + lane;
        if (m < N1) cs_part[(size_t)cs_rem * N1 + m] = accs[0][0];      // this tile's share; tn_colsum_fold_kernel adds the row's shares in order
    }
    // split-K partials leave through LDS (same staging as gemm_nt2's epilogue): the MFMA layout would store 16 B
    // per row per instruction; row-major, one wave instruction writes two whole 512-B rows.
    {
        constexpr int LDS_BYTES = STAGES * STAGE;
        constexpr int CR0 = LDS_BYTES / (BN * 4);
        constexpr int CHUNK = CR0 >= BM ? BM : (CR0 / 32) * 32;
        constexpr int NCH = (BM + CHUNK - 1) / CHUNK;
        constexpr int LPR = BN / 8, RPI = 64 / LPR;
        static_assert(LPR <= 64 && 64 % LPR == 0, "row mapping");
        float* sC = (float*)smem;
        const int u = lane % LPR, n = n0 + u * 8;
        const bool hi = n + 4 < N2;              // N2 % 8 == 0 is checked by the launcher; kept for symmetry
        for (int ch = 0; ch < NCH; ++ch) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rl = wr * TM + i * 32 + (lane & 31);
                if (rl / CHUNK != ch) continue;
                const int r = rl - ch * CHUNK;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int unit = (wc * TN + j * 32 + 8 * g) / 4 + (lane >> 5);
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *(f32x4*)(sC + r * BN + (stage_slot<BN>(unit, r) << 2)) = v;
                    }
            }
            __syncthreads();
            for (int rr = wid * RPI + lane / LPR; rr < CHUNK; rr += NW * RPI) {
                const int m = m0 + ch * CHUNK + rr;
                if (m >= N1 || n >= N2) continue;
                const f32x4 v0 = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u, rr) << 2));
                const f32x4 v1 = *(const f32x4*)(sC + rr * BN + (stage_slot<BN>(2 * u + 1, rr) << 2));
                float* o = Cz + (size_t)m * ldc + n;
                *(f32x4*)o = v0;
                if (hi) *(f32x4*)(o + 4) = v1;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int STAGES, int SCHED>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN) / 4) void gemm_tn2_kernel(const bf16_t* __restrict__ A_,
                                                                            const bf16_t* __restrict__ B_, int Kr,
                                                                            int N1_, int N2_, int lda_, int ldb_,
                                                                            float* __restrict__ C_, int ldc_,
                                                                            size_t split_stride_, int k_chunk,
                                                                            float* __restrict__ colsum_out_,
                                                                            TnSecond g2) {
    // The dispatcher deals the workgroups of a 2-D grid to the XCDs by their LINEAR id (x + y * gridDim.x): the remap has to
    // work on that id.  Logical order = split-major: the workgroups of one XCD at one time share a k range, so the A
    // columns of an output row block and the B columns of an output column block are fetched into that XCD's L2 once.
    const int lin_id = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int split_z = lin_id / (int)gridDim.x;
    const int bid_all = lin_id - split_z * (int)gridDim.x;
    const bool second = bid_all >= g2.tiles1;              // block-uniform
    const bf16_t* A = second ? g2.A : A_;
    const bf16_t* B = second ? g2.B : B_;
    const int N1 = second ? g2.N1 : N1_, N2 = second ? g2.N2 : N2_;
    const int lda = second ? g2.lda : lda_, ldb = second ? g2.ldb : ldb_;
    float* C = second ? g2.C : C_;
    const int ldc = second ? g2.ldc : ldc_;
    const size_t split_stride = second ? g2.split_stride : split_stride_;
    float* colsum_out = second ? g2.colsum_out : colsum_out_;
    const int tiles_n = (N2 + BN - 1) / BN;
    const int bid = second ? bid_all - g2.tiles1 : bid_all;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int kb = split_z * k_chunk;
    const int ke = min(Kr, kb + k_chunk);
    tn2_tile<BM, BN, WM, WN, STAGES, SCHED>(A, B, Kr, N1, N2, lda, ldb, C + (size_t)split_z * split_stride, ldc,
                                            colsum_out ? colsum_out + (size_t)split_z * N1 : nullptr, m0, n0, kb, ke);
}

// ------------------------------------------------------------------------- grouped weight gradients, no split-K
// Up to TN_GROUP_MAX problems C_p = A_p^T B_p over the SAME reduction length Kr in ONE launch.  The tiles of all
// problems are numbered consecutively; the first `full` of them (a multiple of the CU count) are computed whole -
// one workgroup loops over all Kr rows and writes its tile of the FINAL matrix (and the column sums of A) - and only
// the `tail` tiles that would leave most CUs idle in a last round are split `tail_splits` ways over the reduction
// into fp32 slabs that tn_tail_reduce_kernel folds.  Against one split-K launch per problem (36 tiles x 7 splits, 64 MB of
// slabs written and read back per launch, 3 launches + 4 reductions per transformer layer) a whole backward pass becomes one
// launch whose k loops are 308 k tiles long.
struct TnGroup {
    TnProblem p[TN_GROUP_MAX];
    int tile_begin[TN_GROUP_MAX + 1];   // prefix sums of the 256x256 tile counts
    int n;
    int full, tail, tail_splits, k_chunk;   // tiles [0, full) unsplit; [full, full + tail) split over the reduction
    float* slabs;                            // [tail_splits][tail][256*256] partial tiles, then [tail_splits][tail][256] column sums
    uint32_t* pace;                          // k pacing counters, one 128-byte line per (XCD, round) of whole tiles; NULL = off
    int pace_lag;                            // phases (of 2 k tiles) a member may run ahead of the slowest one
    float* cs_part;                          // shared column sums: per problem [tiles_n][N1] partial sums at cs_off[p]; NULL = off
    int cs_off[TN_GROUP_MAX];
};

// adds the per-tile shares of the column sums in tile order (grid: problems x 256-column blocks); tile rows that are not shared
// (a tile of the row is a split tail tile) were written by their n0 == 0 tile / the tail reduction and are skipped
__global__ void tn_colsum_fold_kernel(const TnGroup g) {
    const int pi = blockIdx.x;
    const TnProblem& P = g.p[pi];
    if (!P.colsum) return;
    const int tiles_n = (P.N2 + 255) / 256;
    const float* part = g.cs_part + g.cs_off[pi];
    for (int m = blockIdx.y * blockDim.x + threadIdx.x; m < P.N1; m += gridDim.y * blockDim.x) {
        const int row_last = g.tile_begin[pi] + (m / 256) * tiles_n + tiles_n - 1;
        if (row_last >= g.full) continue;
        float s = 0.f;
#pragma unroll 4
        for (int j = 0; j < tiles_n; ++j) s += part[(size_t)j * P.N1 + m];
        P.colsum[m] = s;
    }
}

__device__ __forceinline__ int tn_group_find(const TnGroup& g, int tile) {
    int p = 0;
    while (p + 1 < g.n && tile >= g.tile_begin[p + 1]) ++p;
    return p;
}

template <int BM, int BN, int WM, int WN, int STAGES, int SCHED>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN) / 4) void gemm_tn2_group_kernel(const TnGroup g, int Kr) {
    // XCD-aware order, applied to the two ranges SEPARATELY (g.full is a multiple of the CU count, so a block's XCD is
    // blockIdx % 8 in both): whole tiles and tail slices differ in length by the split factor, and remapping the grid as
    // one range would hand some XCDs only slices and others a round more of whole tiles.
    const bool whole = (int)blockIdx.x < g.full;
    int tile, z = 0;
    if (whole) {
        tile = xcd_remap(blockIdx.x, g.full);
    } else {                                                // split-major: one XCD works on one k range at a time
        const int t = xcd_remap((int)blockIdx.x - g.full, (int)gridDim.x - g.full);
        z = t / g.tail;
        tile = g.full + (t - z * g.tail);
    }
    const int pi = tn_group_find(g, tile);
    const TnProblem& P = g.p[pi];
    const int tiles_n = (P.N2 + BN - 1) / BN;
    const int bid = tile - g.tile_begin[pi];
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    if (whole) {
        // pace group = the 32 whole tiles one XCD runs in one round (block b sits on XCD b % 8, in dispatch order)
        uint32_t* pace = g.pace ? g.pace + (size_t)((((int)blockIdx.x >> 3) / 32) * 8 + ((int)blockIdx.x & 7)) * 32 : nullptr;
        // the tile row's column sums are shared by its tiles when all of them are whole tiles
        const int row_last = g.tile_begin[pi] + (bid / tiles_n) * tiles_n + tiles_n - 1;
        const bool spread = g.cs_part != nullptr && P.colsum != nullptr && row_last < g.full;
        tn2_tile<BM, BN, WM, WN, STAGES, SCHED>(P.A, P.B, Kr, P.N1, P.N2, P.lda, P.ldb, P.C, P.ldc, P.colsum, m0, n0, 0, Kr, pace,
                                                32u, spread ? tiles_n : 0, spread ? bid % tiles_n : 0, g.pace_lag,
                                                spread ? g.cs_part + g.cs_off[pi] : nullptr);
    } else {
        // slab of (z, tail tile): a dense [BM][BN] tile; the pointers are biased so that the tile's own (m0, n0) indexing of
        // a [*, BN] matrix lands in it (never dereferenced outside the slab: rows / columns beyond N1 / N2 are masked)
        const int ti = tile - g.full;
        float* slab = g.slabs + ((size_t)z * g.tail + ti) * (size_t)(BM * BN);
        float* cs = P.colsum ? g.slabs + (size_t)g.tail_splits * g.tail * (size_t)(BM * BN) + ((size_t)z * g.tail + ti) * BM
                             : nullptr;
        const int kb = z * g.k_chunk, ke = min(Kr, kb + g.k_chunk);
        tn2_tile<BM, BN, WM, WN, STAGES, SCHED>(P.A, P.B, Kr, P.N1, P.N2, P.lda, P.ldb, slab - ((ptrdiff_t)m0 * BN + n0), BN,
                                                cs ? cs - m0 : nullptr, m0, n0, kb, ke);
    }
}

// folds the split slabs of the tail tiles into the final matrices (and the column sums): one workgroup per (tile, 8 rows) -
// 32 x tail workgroups, each wave streams two 1-KB rows of every slab (HBM bound: 64 MB of slabs in the worst case)
__global__ void tn_tail_reduce_kernel(const TnGroup g) {
    constexpr int BM = 256, BN = 256;
    const int ti = blockIdx.x, rb = blockIdx.y;            // tail tile, 8-row block
    const int tile = g.full + ti;
    const int pi = tn_group_find(g, tile);
    const TnProblem& P = g.p[pi];
    const int tiles_n = (P.N2 + BN - 1) / BN;
    const int bid = tile - g.tile_begin[pi];
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int c4 = (threadIdx.x & 63) * 4, r0 = rb * 8 + (threadIdx.x >> 6) * 2;
    const size_t slab = (size_t)(BM * BN), zs = (size_t)g.tail * slab;
    const float* base = g.slabs + (size_t)ti * slab + (size_t)r0 * BN + c4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < g.tail_splits; ++z) {
        s0 += *(const f32x4*)(base + (size_t)z * zs);
        s1 += *(const f32x4*)(base + (size_t)z * zs + BN);
    }
    if (n0 + c4 < P.N2) {
        if (m0 + r0 < P.N1) *(f32x4*)(P.C + (size_t)(m0 + r0) * P.ldc + n0 + c4) = s0;
        if (m0 + r0 + 1 < P.N1) *(f32x4*)(P.C + (size_t)(m0 + r0 + 1) * P.ldc + n0 + c4) = s1;
    }
    if (P.colsum && n0 == 0 && rb == 0) {
        const int m = threadIdx.x;
        if (m < BM && m0 + m < P.N1) {
            float s = 0.f;
            const float* cs = g.slabs + (size_t)g.tail_splits * g.tail * slab;
            for (int z = 0; z < g.tail_splits; ++z) s += cs[((size_t)z * g.tail + ti) * BM + m];
            P.colsum[m0 + m] = s;
        }
    }
}

// also folds the per-split column sums (bias gradient) when cs_ws != nullptr
__global__ void splitk_reduce2_kernel(const float* __restrict__ ws, int splits, int rows, int cols,
                                      float* __restrict__ out, int ldo, float alpha, int accumulate,
                                      const float* __restrict__ cs_ws, float* __restrict__ cs_out) {
    const int c4 = cols >> 2;
    const size_t total = (size_t)rows * c4;
    if (cs_ws) {
        const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (t < (size_t)rows) {
            float s = 0.f;
            for (int z = 0; z < splits; ++z) s += cs_ws[(size_t)z * rows + t];
            cs_out[t] = s;
        }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c4), c = (int)(i % c4) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < splits; ++z) s += *(const f32x4*)(ws + ((size_t)z * rows + r) * cols + c);
        s *= alpha;
        float* o = out + (size_t)r * ldo + c;
        if (accumulate) s += *(const f32x4*)o;
        *(f32x4*)o = s;
    }
}

static int tn2_splits(int Kr, int N1, int N2, int BM, int BN) {
    const int tiles = ((N1 + BM - 1) / BM) * ((N2 + BN - 1) / BN);
    const int ktiles = (Kr + BK2 - 1) / BK2;
    int s = 256 / tiles;
    if (s < 1) s = 1;
    const int max_s = (ktiles + 5) / 6;   // >= ~6 k-tiles per split keeps the pipe busy
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

static void tn2_tile(int* BM, int* BN) {
    if (gemm_cfg() == 1) { *BM = 256; *BN = 128; }
    else { *BM = 256; *BN = 256; }
}

size_t gemm_tn2_workspace_bytes(int Kr, int N1, int N2) {
    // the largest split count over the configurations (bigger tiles -> fewer tiles -> more splits)
    const int s1 = tn2_splits(Kr, N1, N2, 256, 128), s2 = tn2_splits(Kr, N1, N2, 256, 256);
    const int s = s1 > s2 ? s1 : s2;
    return ((size_t)s * N1 * N2 + (size_t)s * N1) * sizeof(float);
}

template <int BM, int BN, int WM, int WN, int STAGES, int SCHED = 0>
static int launch_tn2(const bf16_t* A, const bf16_t* B, int Kr, int N1, int N2, int lda, int ldb, float* ws, int splits,
                      int k_chunk, float* cs_ws, hipStream_t st, const TnSecond* second = nullptr) {
    constexpr int LDS = STAGES * (BM + BN) * 128;
    auto kern = gemm_tn2_kernel<BM, BN, WM, WN, STAGES, SCHED>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    int tiles = ((N1 + BM - 1) / BM) * ((N2 + BN - 1) / BN);
    TnSecond g2{};
    g2.tiles1 = INT32_MAX;
    if (second) {
        g2 = *second;
        g2.tiles1 = tiles;
        tiles += ((g2.N1 + BM - 1) / BM) * ((g2.N2 + BN - 1) / BN);
    }
    hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(WM * WN * 64), LDS, st, A, B, Kr, N1, N2, lda, ldb, ws, N2,
                       (size_t)N1 * N2, k_chunk, cs_ws, g2);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// colsum_out (optional, [N1]): column sums of A over all Kr rows (overwritten)
int gemm_tn2(const bf16_t* A, const bf16_t* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
             float alpha, int accumulate, float* colsum_out, float* ws, size_t ws_bytes, hipStream_t st) {
    if (Kr <= 0 || N1 <= 0 || N2 <= 0) return SPN_ERR_ARG;
    if (N1 % 8 || N2 % 8 || lda % 8 || ldb % 8 || ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)Kr * lda * 2 >= (1ull << 32) || (uint64_t)Kr * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (ws_bytes < gemm_tn2_workspace_bytes(Kr, N1, N2)) return SPN_ERR_WORKSPACE;
    int BM, BN;
    tn2_tile(&BM, &BN);
    int splits = tn2_splits(Kr, N1, N2, BM, BN);
    const int ktiles = (Kr + BK2 - 1) / BK2;
    const int k_chunk = ((ktiles + splits - 1) / splits) * BK2;
    splits = (Kr + k_chunk - 1) / k_chunk;
    float* cs_ws = colsum_out ? ws + (size_t)splits * N1 * N2 : nullptr;
    int rc;
    {
        ProfScope prof(PK_GEMM_TN, 2.0 * Kr * N1 * N2, st);
        switch (gemm_cfg()) {
            case 1: rc = launch_tn2<256, 128, 4, 2, 3>(A, B, Kr, N1, N2, lda, ldb, ws, splits, k_chunk, cs_ws, st); break;
            case 3:
                rc = tn_phased() ? launch_tn2<256, 256, 2, 4, 2, 2>(A, B, Kr, N1, N2, lda, ldb, ws, splits, k_chunk, cs_ws, st)
                                 : launch_tn2<256, 256, 2, 4, 2>(A, B, Kr, N1, N2, lda, ldb, ws, splits, k_chunk, cs_ws, st);
                break;
            default: rc = launch_tn2<256, 256, 4, 2, 2>(A, B, Kr, N1, N2, lda, ldb, ws, splits, k_chunk, cs_ws, st); break;
        }
    }
    if (rc) return rc;
    const size_t total = (size_t)N1 * (N2 / 4);
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if ((size_t)blocks * 256 < (size_t)N1) return SPN_ERR_SHAPE;   // the fused colsum fold needs >= N1 threads
    hipLaunchKernelGGL(splitk_reduce2_kernel, dim3(blocks), dim3(256), 0, st, ws, splits, N1, N2, C, ldc, alpha, accumulate,
                       (const float*)cs_ws, colsum_out);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int device_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

static constexpr size_t TN_PACE_BYTES = 64 * 1024;          // pacing counters: 128 B per (XCD, round), <= 512 groups
static constexpr size_t TN_CS_BYTES = 4 * 1024 * 1024;      // shared column sums: per-tile partial sums of a grouped launch

// SPN_TN_PACE=1 turns the k pacing of the grouped launch on.  Default OFF, measured (round 3, same box, bench step):
// without 13.32 ms / step, grouped launch 1 243 TFLOP/s; with 14.29 ms, 882 TFLOP/s.  The tiles of a group are not equally
// fast - the n0 == 0 tiles carry the bias-gradient MFMAs (+25 % matrix work) - and pacing holds all 32 to the slowest; what
// the shared panels save in L2 misses is less than what the fast tiles lose waiting.
static bool tn_pace_on() {
    static const bool on = [] {
        const char* e = spn_env("SPN_TN_PACE");
        return e && e[0] == '1';
    }();
    return on;
}

size_t gemm_tn_grouped_workspace_bytes(int Kr) {
    (void)Kr;
    return (size_t)device_cu_count() * (256 * 256 + 256) * sizeof(float) + 256 + TN_PACE_BYTES + TN_CS_BYTES;   // at most one slab per CU
}

int gemm_tn_grouped(const TnProblem* probs, int n, int Kr, float* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0 || n > TN_GROUP_MAX || Kr <= 0 || !probs) return SPN_ERR_ARG;
    TnGroup g{};
    g.n = n;
    int total = 0;
    double flops = 0.0;
    for (int i = 0; i < n; ++i) {
        const TnProblem& P = probs[i];
        if (!P.A || !P.B || !P.C || P.N1 <= 0 || P.N2 <= 0) return SPN_ERR_ARG;
        if (P.N1 % 8 || P.N2 % 8 || P.lda % 8 || P.ldb % 8 || P.ldc % 4) return SPN_ERR_SHAPE;
        if ((uint64_t)Kr * P.lda * 2 >= (1ull << 32) || (uint64_t)Kr * P.ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
        g.p[i] = P;
        g.tile_begin[i] = total;
        total += ((P.N1 + 255) / 256) * ((P.N2 + 255) / 256);
        flops += 2.0 * Kr * P.N1 * P.N2;
    }
    g.tile_begin[n] = total;
    const int cus = device_cu_count();
    const int ktiles = (Kr + BK2 - 1) / BK2;
    int tail = total % cus, splits = 1;
    if (tail > 0) {
        splits = cus / tail;                              // fill one round with the tail's slices
        const int max_s = (ktiles + 5) / 6;               // >= ~6 k tiles per slice
        if (splits > max_s) splits = max_s;
        if (splits < 1) splits = 1;
    }
    if (splits <= 1) tail = 0;                            // a tail of more than half a round runs unsplit
    g.full = total - tail;
    g.tail = tail;
    g.k_chunk = ((ktiles + splits - 1) / splits) * BK2;
    g.tail_splits = tail ? (Kr + g.k_chunk - 1) / g.k_chunk : 1;
    g.slabs = ws;
    const size_t slab_bytes = tail ? (size_t)g.tail_splits * tail * (256 * 256 + 256) * sizeof(float) : 0;
    if (tail && (!ws || ws_bytes < slab_bytes)) return SPN_ERR_WORKSPACE;
    // k pacing of the whole tiles: needs every round of an XCD to be 32 tiles (full is a multiple of the CU count), at least
    // two of them sharing panels to be worth it, and room for the counters behind the slabs
    g.pace = nullptr;
    static const int pace_lag = env_int_min1("SPN_TN_PACE_LAG", 1) - 1;      // SPN_TN_PACE_LAG = lag + 1
    g.pace_lag = pace_lag;
    const int groups = (g.full / cus) * 8;
    if (tn_pace_on() && cus == 256 && g.full >= cus && ws && ws_bytes >= slab_bytes + TN_PACE_BYTES &&
        (size_t)groups * 128 <= TN_PACE_BYTES) {
        g.pace = (uint32_t*)((char*)ws + ((slab_bytes + 255) & ~(size_t)255));
        if (hipMemsetAsync(g.pace, 0, (size_t)groups * 128, st) != hipSuccess) g.pace = nullptr;
    }
    // shared column sums (SPN_TN_CS_SPREAD=0: the n0 == 0 tile of every row computes them alone, as in round 2): partial
    // sums [tiles_n][N1] per problem behind the slabs and the pacing counters
    static const bool cs_spread = [] {
        const char* e = spn_env("SPN_TN_CS_SPREAD");
        return !(e && e[0] == '0');
    }();
    g.cs_part = nullptr;
    if (cs_spread && ws) {
        size_t need = 0;
        bool any = false;
        for (int i = 0; i < n; ++i) {
            g.cs_off[i] = (int)need;
            if (probs[i].colsum) {
                any = true;
                need += (size_t)((probs[i].N2 + 255) / 256) * probs[i].N1;
            }
        }
        const size_t base = ((slab_bytes + 255) & ~(size_t)255) + TN_PACE_BYTES;
        if (any && need * sizeof(float) <= TN_CS_BYTES && ws_bytes >= base + need * sizeof(float))
            g.cs_part = (float*)((char*)ws + base);
    }
    constexpr int LDS = 2 * (256 + 256) * 128;
    auto kern = gemm_tn2_group_kernel<256, 256, 2, 4, 2, 2>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        ProfScope prof(PK_GEMM_TN, flops, st);
        hipLaunchKernelGGL(kern, dim3(g.full + g.tail * g.tail_splits), dim3(512), LDS, st, g, Kr);
        SPN_CHECK_LAUNCH();
        if (tail) {
            hipLaunchKernelGGL(tn_tail_reduce_kernel, dim3(tail, 32), dim3(256), 0, st, g);
            SPN_CHECK_LAUNCH();
        }
        if (g.cs_part) {
            hipLaunchKernelGGL(tn_colsum_fold_kernel, dim3(n, 16), dim3(256), 0, st, g);
            SPN_CHECK_LAUNCH();
        }
    }
    return SPN_OK;
}

// Two weight gradients over the same Kr rows in ONE launch (+ one reduction each): C1 = A1^T B1, C2 = A2^T B2, both
// overwritten (alpha 1), optional column sums.  The split count is chosen for the tiles of both, so the small problem
// costs its share of a full wave of blocks instead of a launch of its own (768x768 next to 2304x768: 146 -> 125 us).
size_t gemm_tn2_pair_workspace_bytes(int Kr, int N1a, int N2a, int N1b, int N2b) {
    const int tiles = ((N1a + 255) / 256) * ((N2a + 255) / 256) + ((N1b + 255) / 256) * ((N2b + 255) / 256);
    const int ktiles = (Kr + BK2 - 1) / BK2;
    int s = 256 / tiles;
    const int max_s = (ktiles + 5) / 6;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return (size_t)s * ((size_t)N1a * N2a + (size_t)N1b * N2b + N1a + N1b) * sizeof(float);
}

// pairing pays when the larger problem runs on this kernel anyway (gemm_tn's routing rule) and nothing was overridden;
// SPN_TN_PAIR=0 keeps the two launches (A/B switch)
bool gemm_tn2_pair_ok(int N1a, int N2a) {
    static const bool off = [] {
        const char* e = spn_env("SPN_TN_PAIR");
        return e && e[0] == '0';
    }();
    return !off && gemm_cfg() == 3 && tn_phased() && (size_t)N1a * N2a >= (size_t)768 * 2304;
}

int gemm_tn2_pair(const bf16_t* A1, const bf16_t* B1, int N1a, int N2a, int lda1, int ldb1, float* C1, int ldc1, float* cs1,
                  const bf16_t* A2, const bf16_t* B2, int N1b, int N2b, int lda2, int ldb2, float* C2, int ldc2, float* cs2,
                  int Kr, float* ws, size_t ws_bytes, hipStream_t st) {
    if (Kr <= 0 || N1a <= 0 || N2a <= 0 || N1b <= 0 || N2b <= 0) return SPN_ERR_ARG;
    if (N1a % 8 || N2a % 8 || N1b % 8 || N2b % 8 || lda1 % 8 || ldb1 % 8 || lda2 % 8 || ldb2 % 8 || ldc1 % 4 || ldc2 % 4)
        return SPN_ERR_SHAPE;
    const uint64_t lim = 1ull << 32;
    if ((uint64_t)Kr * lda1 * 2 >= lim || (uint64_t)Kr * ldb1 * 2 >= lim || (uint64_t)Kr * lda2 * 2 >= lim ||
        (uint64_t)Kr * ldb2 * 2 >= lim)
        return SPN_ERR_SHAPE;
    if (gemm_cfg() != 3 || !tn_phased()) return SPN_ERR_ARG;          // only the default 256x256 staggered kernel is grouped
    if (ws_bytes < gemm_tn2_pair_workspace_bytes(Kr, N1a, N2a, N1b, N2b)) return SPN_ERR_WORKSPACE;
    const int tiles = ((N1a + 255) / 256) * ((N2a + 255) / 256) + ((N1b + 255) / 256) * ((N2b + 255) / 256);
    const int ktiles = (Kr + BK2 - 1) / BK2;
    int splits = 256 / tiles;
    const int max_s = (ktiles + 5) / 6;
    if (splits > max_s) splits = max_s;
    if (splits < 1) splits = 1;
    const int k_chunk = ((ktiles + splits - 1) / splits) * BK2;
    splits = (Kr + k_chunk - 1) / k_chunk;
    float* p1 = ws;
    float* p2 = p1 + (size_t)splits * N1a * N2a;
    float* c1 = p2 + (size_t)splits * N1b * N2b;
    float* c2 = c1 + (size_t)splits * N1a;
    TnSecond g2{};
    g2.A = A2; g2.B = B2; g2.N1 = N1b; g2.N2 = N2b; g2.lda = lda2; g2.ldb = ldb2;
    g2.C = p2; g2.ldc = N2b; g2.split_stride = (size_t)N1b * N2b; g2.colsum_out = cs2 ? c2 : nullptr;
    int rc;
    {
        ProfScope prof(PK_GEMM_TN, 2.0 * Kr * ((double)N1a * N2a + (double)N1b * N2b), st);
        rc = launch_tn2<256, 256, 2, 4, 2, 2>(A1, B1, Kr, N1a, N2a, lda1, ldb1, p1, splits, k_chunk, cs1 ? c1 : nullptr, st, &g2);
    }
    if (rc) return rc;
    auto reduce = [&](const float* part, int N1, int N2, float* C, int ldc, const float* csw, float* cs) -> int {
        const size_t total = (size_t)N1 * (N2 / 4);
        const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        if ((size_t)blocks * 256 < (size_t)N1) return SPN_ERR_SHAPE;
        hipLaunchKernelGGL(splitk_reduce2_kernel, dim3(blocks), dim3(256), 0, st, part, splits, N1, N2, C, ldc, 1.0f, 0, csw, cs);
        SPN_CHECK_LAUNCH();
        return SPN_OK;
    };
    rc = reduce(p1, N1a, N2a, C1, ldc1, cs1 ? c1 : nullptr, cs1);
    if (rc) return rc;
    return reduce(p2, N1b, N2b, C2, ldc2, cs2 ? c2 : nullptr, cs2);
}

}  // namespace spn
