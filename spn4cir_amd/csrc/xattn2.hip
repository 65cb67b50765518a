// Second generation of the three large products of the absorbed cross-attention (xattn.hip explains the algebra):
//     P  = softmax(Q' X^T)                       xattn2_rows_kernel<.., 0>
//     dS = P o (dO' X^T - delta)                 xattn2_rows_kernel<.., 1>
//     O' = P X,  dQ' = dS X                      xattn2_apply_kernel
// Why a rebuild (round 6): the first generation runs 128 x 128 x 64 tiles on 4-wave workgroups with 32 x 64 (scores) / 64 x 64 wave
// tiles: a k step is so short that the barrier, the request issue and the fragment latency do not amortise, and a workgroup streams
// the sample's X once per 64 rows (rocprof SQ pass, profiles/r05_blip_packed_sq_pmc.txt: matrix pipe busy 0.14 / 0.23 / 0.35 at
// 4.6-6.7 resident waves).  Here ONE 8-wave workgroup per CU owns a tile of 96 rows x all 640 key columns (rows kernel; 2 x 4 waves of
// 48 x 80 on each of two column halves, both halves' accumulators in registers) or 192 x 384 (apply; 2 x 4 waves of 96 x 96): 30-36
// MFMA per wave between barriers, LDS-DMA stages behind counted waits with the requests handed out one per MFMA group, and the next
// phase's fragments prefetched into a second register set behind the current phase's MFMAs.  96 rows = 8 tokens x 12 heads: 4 tiles
// per dense sample at L = 32 -> 512 workgroups = two exact rounds of 256 CUs.  What the kernels are bound by afterwards (bytes and the
// CU's operand path, not the matrix pipe): LABNOTES.md 9.1; tools/x2probe (in-kernel phase stamps, -DX2_ELIM elimination builds).
//
// Operand images in LDS: rows kernel, k step of 64 bf16 = 128-byte rows: row r at r * 128, logical 16-byte k chunk c at position
// c ^ ((r >> 1) & 7) (nt_swz of gemm_v1_tiles.h); apply, k step of 32 (64-byte rows): chunk c at c ^ ((r >> 2) & 3) (nt2_swz<32> of
// gemm2.hip); reduction-major X (apply): panels of [32 k][128 n], the tn image of gemm_v1_tiles.h.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "gemm_v1_tiles.h"

#include <type_traits>

namespace spn {

static constexpr int X2_THREADS = 512, X2_WAVES = 8, X2_BK = 32;

// tools/x2probe: phase stamps (shader clock) of every workgroup, kept in LDS and dumped at the end - probe builds only
#ifndef X2_ELIM
#define X2_ELIM 0        // probe builds only (tools/x2probe): 1 = no MFMA in the k loop, 2 = fragments read once, 4 = no DMA in the k loop (wrong results)
#endif
#ifdef X2_PROBE
__device__ uint64_t* x2_probe_buf;
#define X2_PROBE_SLOTS 192
#define X2_STAMP(i) do { if ((tid & 63) == 0 && (tid == 0 || tid == 448)) ((uint64_t*)(smem + X2_PROBE_OFF))[(i) + (tid ? 96 : 0)] = __builtin_readcyclecounter(); } while (0)
#define X2_DUMP() do { __syncthreads(); if (tid < X2_PROBE_SLOTS) x2_probe_buf[(size_t)blockIdx.x * X2_PROBE_SLOTS + tid] = ((uint64_t*)(smem + X2_PROBE_OFF))[tid]; } while (0)
#else
#define X2_STAMP(i)
#define X2_DUMP()
#endif

__device__ __forceinline__ int x2_swz(int r, int c) { return c ^ ((r >> 2) & 3); }

// byte offset (inside a [rows][32 k] image) of the fragment chunk a lane reads: row r, k chunk c
__device__ __forceinline__ uint32_t x2_frag_off(int r, int c) { return (uint32_t)(r * 64 + (x2_swz(r, c) << 4)); }

__device__ __forceinline__ void glds16_s(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, uint32_t voffset_bytes, uint32_t soffset_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_base), 16, voffset_bytes, soffset_bytes, 0, SPN_GLDS_AUX);
}

// ----------------------------------------------------------------------------------------------------------------------------
// rows kernel.  A workgroup owns 96 rows of one sample and ALL NH * NJ * 64 key columns, in NH column halves of HC = NJ * 64 walked
// one after the other (the accumulators of both halves stay in registers); 2 x 4 waves, wave tile 48 rows x NJ * 16 columns of a
// half.  k steps of 64 (128-byte rows: the 8 x 128 B LDS-DMA piece is the fast shape of the texture-address path - 55-61 B / clk /
// CU against 28-47 for 16 x 64 B, LABNOTES.md 5.1; the 32-deep first version of this kernel ran AT that limit, k step = 46 KB /
// 28 B per clock, with the matrix pipe waiting).  Three stages of (96 + HC) x 128 B.
// A k step is two MFMA-k phases of 15 MFMA per wave with the fragments of the NEXT phase prefetched behind the current one:
//     phase 1: MFMA(s, kk 0)  ||  read fragments (s, kk 1)           requests: second half of stage s + 2
//     counted vmcnt wait (stage s + 1 landed), lgkmcnt(0), barrier   -> buffer of stage s is free, stage s + 1 is visible
//     phase 2: MFMA(s, kk 1)  ||  read fragments (s + 1, kk 0)       requests: first half of stage s + 3
// and the 6-7 LDS-DMA requests of a wave and stage are handed out one per MFMA group, not as a burst behind the barrier.
// MODE 0: P = softmax over the row (columns >= S masked, written as zero); MODE 1: dS = Pin o (acc - delta[row]).
// ----------------------------------------------------------------------------------------------------------------------------
template <int NJ, int NH, int MODE>
__global__ __launch_bounds__(X2_THREADS, 1) void xattn2_rows_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ X,
                                                                   const bf16_t* __restrict__ Pin, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ out, int R, int S, int E,
                                                                   const int32_t* __restrict__ cu, int H) {
    constexpr int MI = 3, BMR = 96, HC = NJ * 64, SPW = NH * HC;
    constexpr int GA = BMR / 8, GB = HC / 8, G = GA + GB;                // 1 KB DMA requests per stage (8 rows x 128 B each)
    constexpr int A_BYTES = BMR * 128, STAGE = A_BYTES + HC * 128;
    constexpr int REQ_HI = (G + X2_WAVES - 1) / X2_WAVES, REQ_LO = G / X2_WAVES;      // per wave and stage
    constexpr int REQ_P2 = (REQ_HI + 1) / 2;                             // requests 0 .. REQ_P2 - 1 go out in phase 2, the rest in phase 1
    constexpr int PITCH = SPW * 2 + 16;                                  // output image: bf16 rows, 16 B of padding
    static_assert(BMR * PITCH <= 3 * STAGE, "the output image aliases the operand stages");
    static_assert(G % X2_WAVES != 0 && REQ_P2 <= REQ_LO, "vmcnt bookkeeping below assumes REQ_HI = REQ_LO + 1 and a full phase-2 half");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 3 x STAGE, then red[4][BMR] floats
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
#ifdef X2_PROBE
    constexpr int X2_PROBE_OFF = 3 * STAGE + 4 * BMR * 4;
#endif
    X2_STAMP(0);
    const int tiles_r = (R + BMR - 1) / BMR;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = bid / tiles_r, m0 = (bid % tiles_r) * BMR;
    const size_t r0 = cu ? (size_t)cu[b] * H : (size_t)b * R;            // packed: sample b owns the rows cu[b] * H .. cu[b + 1] * H
    if (cu) R = (cu[b + 1] - cu[b]) * H;
    if (m0 >= R) return;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(Q + r0 * E, (uint32_t)R * (uint32_t)E * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(X + (size_t)b * S * E, (uint32_t)S * (uint32_t)E * 2u);
    // DMA plan of this wave: request g = wid + 8 * i covers the image rows 8 g .. 8 g + 7 (A rows first, then the half's X rows)
    const bool more = wid < G % X2_WAVES;                                // these waves issue REQ_HI requests per stage, the others REQ_LO
    uint32_t voff[REQ_HI];
#pragma unroll
    for (int i = 0; i < REQ_HI; ++i) {
        const int g = wid + X2_WAVES * i;
        const int rimg = g * 8 + (lane >> 3);                            // row of the stage image (A rows, then B rows)
        const int c = nt_swz(rimg, lane & 7);                            // (BMR is a multiple of 16: the swizzle phase is the row's own)
        const int grow = g < GA ? m0 + rimg : rimg - BMR;                // X row inside the half
        voff[i] = ((uint32_t)grow * (uint32_t)E + (uint32_t)(c * 8)) * 2u;
    }
    const int nkh = E / 64, NS = NH * nkh;                               // k steps per half / stages in all
    // request i (of this wave) of stage s: k slab s % nkh of the A rows and of the X rows of half s / nkh
    auto stage_one = [&](int i, int s, int buf) {
        const int g = wid + X2_WAVES * i;
        if (i < REQ_LO || more) {
            // the second half walks k BACKWARDS: its first A slabs are the ones the first half read last, so the re-read of the
            // workgroup's 147 KB of A rows finds more of them still in the XCD's L2 (the X stream pushes them out in order)
            const int hf = s >= nkh ? 1 : 0, kt = hf ? 2 * nkh - 1 - s : s;
            // the k offset stays inside a row -> scalar offset; the half's row offset decides which X rows exist (rows >= S read as
            // zero) -> vector offset, where the descriptor's range check certainly sees it
            const uint32_t vo = voff[i] + (g < GA ? 0u : (uint32_t)hf * (uint32_t)HC * (uint32_t)E * 2u);
            glds16_s(g < GA ? rsA : rsB, smem + buf * STAGE + g * 1024, vo, (uint32_t)kt * 128u);
        }
    };
    f32x4 acc[NH][MI][NJ];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment addresses (kk = 0) relative to a stage; kk = 1 is the chunk 4 further: position ^ 4, byte offset ^ 64
    uint32_t offA[MI], offB[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wr * 48 + i * 16 + (lane & 15);
        offA[i] = (uint32_t)(r * 128 + (nt_swz(r, lane >> 4) << 4));
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = wc * NJ * 16 + j * 16 + (lane & 15);
        offB[j] = (uint32_t)(A_BYTES + r * 128 + (nt_swz(r, lane >> 4) << 4));
    }
    bf16x8 fa[2][MI], fb[2][NJ];
    auto read_frags = [&](auto kk_, int buf) {
        constexpr int KK = decltype(kk_)::value;
        const char* s = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[KK][i] = *(const bf16x8*)(s + (offA[i] ^ (KK * 64)));
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[KK][j] = *(const bf16x8*)(s + (offB[j] ^ (KK * 64)));
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    // prologue: stages 0 and 1, and the phase-2 half of stage 2 (its other half goes out in phase 1 of step 0)
#pragma unroll
    for (int i = 0; i < REQ_HI; ++i) stage_one(i, 0, 0);
#pragma unroll
    for (int i = 0; i < REQ_HI; ++i) stage_one(i, 1, 1);
#pragma unroll
    for (int i = 0; i < REQ_P2; ++i) stage_one(i, 2, 2);
    if (more) wait_vmcnt<REQ_HI + REQ_P2>();
    else wait_vmcnt<REQ_LO + REQ_P2>();
    __builtin_amdgcn_s_barrier();
    X2_STAMP(1);
    read_frags(K0{}, 0);
    int bcur = 0, bnext = 1, bprev = 2;                                  // buffers of stages s, s + 1, s + 2 (= s - 1)
    auto step = [&](auto hf_, int s) {
        constexpr int HF = decltype(hf_)::value;
        X2_STAMP(8 + 3 * s);
        // ---- phase 1
        if (!(X2_ELIM & 2) || s == 0) read_frags(K1{}, bcur);
        __builtin_amdgcn_sched_barrier(0);
        const bool st1 = s + 2 < NS && !(X2_ELIM & 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (!(X2_ELIM & 1)) acc[HF][i][j] = mfma16(fb[0][j], fa[0][i], acc[HF][i][j]);
            if (REQ_P2 + i < REQ_HI) {                                   // second half of stage s + 2 -> the buffer stage s - 1 left
                __builtin_amdgcn_sched_barrier(0);
                if (st1) stage_one(REQ_P2 + i, s + 2, bprev);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_assert(REQ_HI - REQ_P2 <= MI && REQ_P2 <= 2 * MI, "requests per phase fit the MFMA groups");
        // stage s + 1 has landed once at most the requests of stage s + 2 are outstanding
        if (st1) {
            if (more) wait_vmcnt<REQ_HI>();
            else wait_vmcnt<REQ_LO>();
        } else {
            wait_vmcnt<0>();
        }
        X2_STAMP(9 + 3 * s);
        __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): this wave's kk = 1 fragments have arrived
        __builtin_amdgcn_s_barrier();                                    // stage s + 1 visible to all; nobody reads buffer bcur any more
        X2_STAMP(10 + 3 * s);
        // ---- phase 2
        if (!(X2_ELIM & 2)) read_frags(K0{}, bnext);                     // (the last step reads a stale buffer and drops it)
        __builtin_amdgcn_sched_barrier(0);
        const bool st2 = s + 3 < NS && !(X2_ELIM & 4);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (!(X2_ELIM & 1)) acc[HF][i][j] = mfma16(fb[1][j], fa[1][i], acc[HF][i][j]);
                if ((i * NJ + j) % 2 == 1 && (i * NJ + j) / 2 < REQ_P2) {    // first half of stage s + 3 -> the buffer of stage s
                    __builtin_amdgcn_sched_barrier(0);
                    if (st2) stage_one((i * NJ + j) / 2, s + 3, bcur);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const int t = bcur;
        bcur = bnext;
        bnext = bprev;
        bprev = t;
    };
    for (int s = 0; s < nkh; ++s) step(K0{}, s);
    if constexpr (NH == 2)
        for (int s = nkh; s < NS; ++s) step(K1{}, s);
    X2_STAMP(2);
    // a lane holds, of row wr * 48 + i * 16 + (lane & 15), the columns h * HC + wc * NJ * 16 + j * 16 + (lane >> 4) * 4 + e
    float* red = (float*)(smem + 3 * STAGE);                             // [4 wave columns][BMR rows]
    char* img = smem;
    const int col0 = wc * NJ * 16 + (lane >> 4) * 4;
    if constexpr (MODE == 0) {
        constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float mx = -INFINITY;
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (h * HC + wc * NJ * 16 + j * 16 + 16 > S) {       // wave-uniform: only the tiles that straddle or pass S mask
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (h * HC + col0 + j * 16 + e >= S) acc[h][i][j][e] = -INFINITY;
                    }
                    mx = fmaxf(fmaxf(mx, fmaxf(acc[h][i][j][0], acc[h][i][j][1])), fmaxf(acc[h][i][j][2], acc[h][i][j][3]));
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (lane < 16) red[wc * BMR + wr * 48 + i * 16 + lane] = mx;
        }
        __syncthreads();
        float rsum[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wr * 48 + i * 16 + (lane & 15);
            const float mx = fmaxf(fmaxf(red[row], red[BMR + row]), fmaxf(red[2 * BMR + row], red[3 * BMR + row]));
            const float nm = -mx * LOG2E;
            float sm = 0.f;
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[h][i][j][e], LOG2E, nm));     // exp(x - max)
                        acc[h][i][j][e] = p;
                        sm += p;
                    }
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            rsum[i] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (lane < 16) red[wc * BMR + wr * 48 + i * 16 + lane] = rsum[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wr * 48 + i * 16 + (lane & 15);
            const float sm = (red[row] + red[BMR + row]) + (red[2 * BMR + row] + red[3 * BMR + row]);   // fixed order: same sum in every wave
            const float inv = 1.0f / sm;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                char* o = img + row * PITCH + (h * HC + col0) * 2;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 v = acc[h][i][j] * inv;
                    *(bf16x4*)(o + j * 32) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                }
            }
        }
    } else {
        __syncthreads();                                                 // every wave has left the k loop: the stages are free
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wr * 48 + i * 16 + (lane & 15), m = m0 + row;
            const bool ok = m < R;
            const float dl = ok ? delta[r0 + m] : 0.f;
            const bf16_t* pr = Pin + (r0 + (ok ? m : m0)) * SPW + col0;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                bf16x4 p4[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) p4[j] = *(const bf16x4*)(pr + h * HC + j * 16);
                char* o = img + row * PITCH + (h * HC + col0) * 2;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = bf2f(p4[j][e]) * (acc[h][i][j][e] - dl);
                    *(bf16x4*)(o + j * 32) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                }
            }
        }
    }
    __syncthreads();
    X2_STAMP(3);
    // whole rows out: 16 bytes per lane, SPW / 8 lanes per row
    constexpr int UPR = SPW / 8;
    const int rows_valid = min(BMR, R - m0);
    bf16_t* dst = out + (r0 + m0) * SPW;
#ifndef X2_STORE_WT
#define X2_STORE_WT 1        // write-through (sc1) output stores: the 63-75 MB output stream does not push the X / A lines the other
#endif                       // workgroups of the sample are about to re-read out of the XCD's L2
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc(dst, (uint32_t)rows_valid * (uint32_t)SPW * 2u);
    for (int u = tid; u < rows_valid * UPR; u += X2_THREADS) {
        const int r = u / UPR, c = u % UPR;
        const u32x4 v = *(const u32x4*)(img + r * PITCH + c * 16);
        if (X2_STORE_WT) store_wt16(rsO, ((size_t)r * SPW + c * 8) * 2, v);
        else *(u32x4*)(dst + (size_t)r * SPW + c * 8) = v;
    }
    X2_STAMP(4);
    X2_DUMP();
}

// ----------------------------------------------------------------------------------------------------------------------------
// apply: out[b, r, :] = A[b, r, 0..SPA) . X[b, 0..S, :]   (A = P or dS, k-contiguous; X reduction-major).  Tile 192 rows x
// NJ * 64 columns (NJ = 6: 384, NJ = 4: 256), waves 2 x 4, wave tile 96 x NJ * 16.  k steps of 32 rows of X, FOUR LDS stages:
// A image [192][32 k] (12 KB) + NJ / 2 panels of [32 k][128 n] (8 KB each).
// A k step is two phases of 3 x NJ MFMA (row tiles 0-2, then 3-5 of the wave) with the next phase's fragments prefetched:
//     phase 1: MFMA rows 0-2 (a0, b)  ||  read a1 = A rows 3-5 of stage s          requests: second half of stage s + 3
//     counted vmcnt wait (stage s + 1 landed), lgkmcnt(0), barrier                 -> buffer of stage s free, stage s + 1 visible
//     phase 2: MFMA rows 3-5 (a1, b)  ||  read b', a0 of stage s + 1               requests: first half of stage s + 4
// (b double-buffered across steps).  Requests go out one per MFMA group, as in the rows kernel.
// ----------------------------------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(X2_THREADS, 1) void xattn2_apply_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ X,
                                                                    bf16_t* __restrict__ out, int R, int S, int E, int SPA,
                                                                    const int32_t* __restrict__ cu, int H) {
    constexpr int BMA = 192, BNA = NJ * 64, PANELS = BNA / 128, NST = 4;
    constexpr int GA = BMA / 16, GB = PANELS * 8, G = GA + GB;           // 1 KB DMA requests per stage
    constexpr int A_BYTES = BMA * 64, STAGE = A_BYTES + PANELS * 8192;
    constexpr int REQ_HI = (G + X2_WAVES - 1) / X2_WAVES, REQ_LO = G / X2_WAVES;
    constexpr int REQ_P2 = (REQ_HI + 1) / 2;                             // requests 0 .. REQ_P2 - 1 go out in phase 2, the rest in phase 1
    constexpr int PITCH = BNA * 2 + 16;
    static_assert(96 * PITCH <= NST * STAGE, "the output image (one row half at a time) aliases the operand stages");
    static_assert(G % X2_WAVES != 0 && REQ_P2 <= REQ_LO && REQ_P2 <= 3 && REQ_HI - REQ_P2 <= 3, "request bookkeeping below");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // NST x STAGE
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int tiles_n = E / BNA, tiles_m = (R + BMA - 1) / BMA;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (bid % tiles_n) * BNA, m0 = ((bid / tiles_n) % tiles_m) * BMA, b = bid / (tiles_n * tiles_m);
    const size_t r0 = cu ? (size_t)cu[b] * H : (size_t)b * R;
    if (cu) R = (cu[b + 1] - cu[b]) * H;
    if (m0 >= R) return;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + r0 * SPA, (uint32_t)R * (uint32_t)SPA * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(X + (size_t)b * S * E, (uint32_t)S * (uint32_t)E * 2u);
    const bool more = wid < G % X2_WAVES;
    uint32_t voff[REQ_HI];
#pragma unroll
    for (int i = 0; i < REQ_HI; ++i) {
        const int g = wid + X2_WAVES * i;
        if (g < GA) {                                                    // 16 rows of A, 64 B each
            const int r = g * 16 + (lane >> 2);
            voff[i] = ((uint32_t)(m0 + r) * (uint32_t)SPA + (uint32_t)(x2_swz(r, lane & 3) * 8)) * 2u;
        } else {                                                         // 4 k rows of one [32][128] panel, 256 B each (tn_stage)
            const int gb = g - GA, panel = gb >> 3, k = (gb & 7) * 4 + (lane >> 4);
            const int pos16 = lane & 15, c32 = (pos16 >> 1) ^ tn_f(k);
            voff[i] = ((uint32_t)k * (uint32_t)E + (uint32_t)(n0 + panel * 128 + c32 * 16 + (pos16 & 1) * 8)) * 2u;
        }
    }
    auto stage_one = [&](int i, int kt, int buf) {
        const int g = wid + X2_WAVES * i;
        char* dst = smem + buf * STAGE + g * 1024;
        if (i < REQ_LO || more) {
            // A: the k offset stays inside the row (the row clip is in voff) -> scalar offset; X: the k offset IS the row, so it
            // goes through the vector offset, where the descriptor's range check certainly sees it (rows >= S read as zero)
            if (g < GA) glds16_s(rsA, dst, voff[i], (uint32_t)kt * (X2_BK * 2));
            else glds16(rsB, dst, voff[i] + (uint32_t)kt * (uint32_t)(X2_BK * 2) * (uint32_t)E);
        }
    };
    f32x4 acc[6][NJ];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t offA[6], offB[NJ][2];
#pragma unroll
    for (int i = 0; i < 6; ++i) offA[i] = x2_frag_off(wr * 96 + i * 16 + (lane & 15), lane >> 4);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int col = wc * NJ * 16 + j * 16, panel = col >> 7, cb = col & 127;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int krow = (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
            const int p32 = (cb >> 4) ^ tn_f(krow);
            offB[j][h] = A_BYTES + panel * 8192 + krow * 256 + p32 * 32 + (lane & 3) * 8;
        }
    }
    const int nk = SPA / X2_BK;                                          // rows >= S of X read as zero through the descriptor; nk >= 8
    bf16x8 fa[2][3];                                                     // [0]: row tiles 0-2 of a step, [1]: row tiles 3-5
    TnFrag fb[2][NJ];                                                    // b fragments of step parity 0 / 1
    auto read_a = [&](auto half_, int buf) {
        constexpr int HF = decltype(half_)::value;
        const char* s = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i) fa[HF][i] = *(const bf16x8*)(s + offA[HF * 3 + i]);
    };
    auto read_b = [&](auto par_, int buf) {
        constexpr int P = decltype(par_)::value;
        const char* s = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            fb[P][j].h[0] = lds_tr16_b64_asm(s + offB[j][0]);
            fb[P][j].h[1] = lds_tr16_b64_asm(s + offB[j][1]);
        }
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    // prologue: stages 0 .. 2 and the phase-2 half of stage 3
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int i = 0; i < REQ_HI; ++i) stage_one(i, st, st);
#pragma unroll
    for (int i = 0; i < REQ_P2; ++i) stage_one(i, 3, 3);
    if (more) wait_vmcnt<2 * REQ_HI + REQ_P2>();
    else wait_vmcnt<2 * REQ_LO + REQ_P2>();
    __builtin_amdgcn_s_barrier();
    read_b(C0{}, 0);
    read_a(C0{}, 0);
    int b0 = 0, b1 = 1, b3 = 3;                                          // buffers of stages s, s + 1, s + 3 (= s - 1)
    auto step = [&](auto par_, int s) {
        constexpr int P = decltype(par_)::value;
        // ---- phase 1: rows 0-2 against b[P]; a1 of this stage arrives behind them
        read_a(C1{}, b0);
        wait_lgkm<3>();                                                  // everything but the three a1 reads just issued: b[P], a0 are in
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            lds_tie(fb[P][j].h[0]);
            lds_tie(fb[P][j].h[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 bb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bb[j] = tn_tie(fb[P][j]);
        const bool st1 = s + 3 < nk;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(bb[j], fa[0][i], acc[i][j]);
            if (REQ_P2 + i < REQ_HI) {                                   // second half of stage s + 3 -> the buffer stage s - 1 left
                __builtin_amdgcn_sched_barrier(0);
                if (st1) stage_one(REQ_P2 + i, s + 3, b3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // stage s + 1 has landed once at most the requests of stages s + 2 and s + 3 are outstanding
        if (s + 3 < nk) {
            if (more) wait_vmcnt<2 * REQ_HI>();
            else wait_vmcnt<2 * REQ_LO>();
        } else if (s + 2 < nk) {
            if (more) wait_vmcnt<REQ_HI>();
            else wait_vmcnt<REQ_LO>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): a1 has arrived
        __builtin_amdgcn_s_barrier();                                    // stage s + 1 visible to all; nobody reads buffer b0 any more
        // ---- phase 2: rows 3-5 against b[P]; b[1 - P] and a0 of the next stage arrive behind them
        read_b(std::integral_constant<int, 1 - P>{}, b1);                // (the last step reads a stale buffer and drops it)
        read_a(C0{}, b1);
        __builtin_amdgcn_sched_barrier(0);
        const bool st2 = s + 4 < nk;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[3 + i][j] = mfma16(bb[j], fa[1][i], acc[3 + i][j]);
            if (i < REQ_P2) {                                            // first half of stage s + 4 -> the buffer of stage s
                __builtin_amdgcn_sched_barrier(0);
                if (st2) stage_one(i, s + 4, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        b3 = b0;
        b0 = b1;
        b1 = b1 == NST - 1 ? 0 : b1 + 1;
    };
    for (int s = 0; s < nk; s += 2) {                                    // SPA / 32 is even
        step(C0{}, s);
        step(C1{}, s + 1);
    }
    // the two row halves leave one after the other through a [96][BNA] bf16 image: whole 2 * BNA-byte row pieces per instruction
    constexpr int UPR = BNA / 8;
    char* img = smem;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                                 // k loop / the previous half's readers are done with LDS
        if (wr == half) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                char* o = img + (i * 16 + (lane & 15)) * PITCH + (wc * NJ * 16 + (lane >> 4) * 4) * 2;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 v = acc[i][j];
                    *(bf16x4*)(o + j * 32) = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                }
            }
        }
        __syncthreads();
        const int mh = m0 + half * 96;
        const int rows_valid = min(96, R - mh);
        bf16_t* dst = out + (r0 + mh) * E + n0;
        const __amdgpu_buffer_rsrc_t rsO = make_rsrc(dst, rows_valid > 0 ? ((uint32_t)(rows_valid - 1) * (uint32_t)E + (uint32_t)BNA) * 2u : 0u);
        for (int u = tid; u < rows_valid * UPR; u += X2_THREADS) {
            const int r = u / UPR, c = u % UPR;
            const u32x4 v = *(const u32x4*)(img + r * PITCH + c * 16);
            if (X2_STORE_WT) store_wt16(rsO, ((size_t)r * E + c * 8) * 2, v);
            else *(u32x4*)(dst + (size_t)r * E + c * 8) = v;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------------------------------
bool xattn2_on() {
    static const bool on = [] {
        const char* e = spn_env("SPN_XATTN_V2");                  // 0: the first-generation kernels of xattn.hip (A/B switch)
        return !(e && e[0] == '0');
    }();
    return on;
}

template <typename K>
static int x2_lds(K kern, int bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e == hipSuccess ? SPN_OK : (int)e;
}

template <int NJ, int NH, int MODE>
static int launch_rows(const bf16_t* Q, const bf16_t* X, const bf16_t* Pin, const float* delta, bf16_t* out, int B, int R, int S, int E,
                       hipStream_t st, const int32_t* cu, int H) {
    constexpr int BMR = 96, HC = NJ * 64;
#ifdef X2_PROBE
    constexpr int LDS = 3 * (BMR + HC) * 128 + 4 * BMR * 4 + X2_PROBE_SLOTS * 8;
#else
    constexpr int LDS = 3 * (BMR + HC) * 128 + 4 * BMR * 4;
#endif
    static_assert(LDS <= 160 * 1024, "LDS per workgroup");
    static const int rc0 = x2_lds(xattn2_rows_kernel<NJ, NH, MODE>, LDS);
    if (rc0) return rc0;
    const int tiles = B * ((R + BMR - 1) / BMR);
    hipLaunchKernelGGL((xattn2_rows_kernel<NJ, NH, MODE>), dim3(tiles), dim3(X2_THREADS), LDS, st, Q, X, Pin, delta, out, R, S, E, cu, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// SP = 640: two halves of 320 columns (4 wave columns x 80); SP = 256: one half of 256 (4 x 64)
int xattn2_scores_softmax(const bf16_t* Q, const bf16_t* X, bf16_t* P, int B, int R, int S, int E, hipStream_t st, const int32_t* cu,
                          int H) {
    if (xattn_sp(S) == 256) return launch_rows<4, 1, 0>(Q, X, nullptr, nullptr, P, B, R, S, E, st, cu, H);
    return launch_rows<5, 2, 0>(Q, X, nullptr, nullptr, P, B, R, S, E, st, cu, H);
}

int xattn2_dscores(const bf16_t* dO, const bf16_t* X, const bf16_t* P, const float* delta, bf16_t* dS, int B, int R, int S, int E,
                   hipStream_t st, const int32_t* cu, int H) {
    if (xattn_sp(S) == 256) return launch_rows<4, 1, 1>(dO, X, P, delta, dS, B, R, S, E, st, cu, H);
    return launch_rows<5, 2, 1>(dO, X, P, delta, dS, B, R, S, E, st, cu, H);
}

bool xattn2_apply_ok(int E) { return E % 384 == 0 || E % 256 == 0; }

template <int NJ>
static int launch_apply(const bf16_t* A, const bf16_t* X, bf16_t* out, int B, int R, int S, int E, int SPA, hipStream_t st,
                        const int32_t* cu, int H) {
    constexpr int BNA = NJ * 64;
    constexpr int LDS = 4 * (192 * 64 + (BNA / 128) * 8192);
    static const int rc0 = x2_lds(xattn2_apply_kernel<NJ>, LDS);
    if (rc0) return rc0;
    const int tiles = B * ((R + 191) / 192) * (E / BNA);
    hipLaunchKernelGGL((xattn2_apply_kernel<NJ>), dim3(tiles), dim3(X2_THREADS), LDS, st, A, X, out, R, S, E, SPA, cu, H);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int xattn2_apply(const bf16_t* A, const bf16_t* X, bf16_t* out, int B, int R, int S, int E, hipStream_t st, const int32_t* cu, int H) {
    const int SPA = xattn_sp(S);
    if (E % 384 == 0) return launch_apply<6>(A, X, out, B, R, S, E, SPA, st, cu, H);
    return launch_apply<4>(A, X, out, B, R, S, E, SPA, st, cu, H);
}

}  // namespace spn
