// Tile primitives of the 128 x 128 x 64 bf16 MFMA kernels (gemm.hip: gemm_nt / gemm_tn; xattn.hip: the absorbed cross-attention):
// LDS-DMA staging of K-contiguous ("nt") and reduction-major ("tn") operand tiles into XOR-swizzled LDS images and the fragment
// reads for v_mfma_f32_16x16x32_bf16.  Device-only; include after common.h.
#pragma once
#include "common.h"

namespace spn {

static constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
static constexpr int TILE_BYTES = 128 * 64 * 2;   // one operand tile = 16 KiB

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// NT mainloop pieces (shared with the bank kernels through kernels.h is not needed; local)
// ---------------------------------------------------------------------------------------
// LDS image of a [128 rows][64 k] bf16 tile: row r at byte r*128; the 16-byte chunk holding
// logical k-chunk c (8 elements) sits at position c ^ ((r>>1)&7).
__device__ __forceinline__ int nt_swz(int r, int c) { return c ^ ((r >> 1) & 7); }

template <int ITER = 4>   // ITER x 32 rows
__device__ __forceinline__ void nt_stage(__amdgpu_buffer_rsrc_t rs, char* sT, int row0, int ld, int k0,
                                         int wid, int lane) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int R0 = (wid * ITER + i) * 8;
        const int r = R0 + (lane >> 3);
        const int c = nt_swz(r, lane & 7);
        const uint32_t off = ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(k0 + c * 8)) * 2u;
        glds16(rs, sT + R0 * 128, off);
    }
}

__device__ __forceinline__ bf16x8 nt_frag(const char* sT, int r, int c) {
    return *(const bf16x8*)(sT + r * 128 + (nt_swz(r, c) << 4));
}

// ---------------------------------------------------------------------------------------
// Reduction-major tiles.  LDS image of a [64 k][128 n] bf16 tile: row k at byte k*256; the 32-byte chunk holding
// logical columns 16*c..16*c+15 sits at position c ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2, so
// the 8 rows touched by one 32-lane half of a ds_read_b64_tr_b16 hit 8 distinct bank groups.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

__device__ __forceinline__ void tn_stage(__amdgpu_buffer_rsrc_t rs, char* sT, int kbase, int ld, int col0,
                                         int wid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R0 = (wid * 4 + i) * 4;
        const int r = R0 + (lane >> 4);
        const int pos16 = lane & 15;
        const int c32 = (pos16 >> 1) ^ tn_f(r);
        const uint32_t off = ((uint32_t)(kbase + r) * (uint32_t)ld + (uint32_t)(col0 + c32 * 16 + (pos16 & 1) * 8)) * 2u;
        glds16(rs, sT + R0 * 256, off);
    }
}

// fragment for the 16 columns [cb, cb+16) and the 32 k-rows [ks*32, ks*32+32): lane l gets
// column cb + (l&15), k = ks*32 + (l>>4)*8 + 0..7
// The reads are the asm form (common.h): in front of the builtin hipcc puts `s_waitcnt vmcnt(0)` while the NEXT k tile's LDS-DMA is
// in flight (it cannot prove the read does not alias the DMA's destination), which serialised load and MFMA phases.  The caller
// waits with wait_lgkm<0>() and converts with tn_tie().
struct TnFrag { s16x4 h[2]; };
__device__ __forceinline__ TnFrag tn_frag(const char* sT, int cb, int ks, int lane) {
    TnFrag u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int krow = ks * 32 + (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
        const int p32 = (cb >> 4) ^ tn_f(krow);
        u.h[h] = lds_tr16_b64_asm(sT + krow * 256 + p32 * 32 + (lane & 3) * 8);
    }
    return u;
}
__device__ __forceinline__ bf16x8 tn_tie(TnFrag& f) {
    lds_tie(f.h[0]);
    lds_tie(f.h[1]);
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = f.h[0];
    u.h[1] = f.h[1];
    return u.v;
}

}  // namespace spn
