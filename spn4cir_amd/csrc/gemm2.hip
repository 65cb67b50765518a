// Second-generation bf16 MFMA GEMMs (gfx950): 256x128 output tile, 8 waves (4x2), wave tile
// 64x64 = 2x2 v_mfma_f32_32x32x16_bf16, 64-deep K step, THREE LDS stages filled by
// buffer_load...lds with a COUNTED s_waitcnt vmcnt so that one whole tile stays in flight across
// the (raw) s_barrier: the prefetch distance is two K-steps, which covers HBM latency; the v1
// kernels (gemm.hip) drain vmcnt(0) every step and are kept for A/B runs (SPN_GEMM_V1=1).
//
//   gemm_nt2 : C[M,N]  = epilogue(A[M,K] . B[N,K]^T)
//   gemm_tn2 : C[N1,N2] = A[Kr,N1]^T . B[Kr,N2]  (+ optional column sums of A = bias gradient,
//              obtained for free with an all-ones MFMA operand)
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static constexpr int BM2 = 256, BN2 = 128, BK2 = 64, NT2 = 512;
static constexpr int A2_BYTES = 256 * 64 * 2, B2_BYTES = 128 * 64 * 2, STAGE2 = A2_BYTES + B2_BYTES;   // 48 KiB
static constexpr int LDS2 = 3 * STAGE2;                                                                 // 144 KiB
static constexpr int GLDS_PER_STAGE = 6;   // per wave: 4 (A) + 2 (B)

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

// ----------------------------------------------------------------------------------- NT
// LDS image of a [rows][64 k] bf16 tile: row r at byte r*128, logical 16-B k-chunk c at
// position c ^ ((r>>1)&7) (conflict-free for the 32-row ds_read_b128 fragments).
__device__ __forceinline__ int nt2_swz(int r, int c) { return c ^ ((r >> 1) & 7); }

template <int PER_WAVE>
__device__ __forceinline__ void nt2_stage(__amdgpu_buffer_rsrc_t rs, char* sT, int row0, int ld, int k0, int wid,
                                          int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int R0 = (wid * PER_WAVE + i) * 8;
        const int r = R0 + (lane >> 3);
        const int c = nt2_swz(r, lane & 7);
        glds16(rs, sT + R0 * 128, ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(k0 + c * 8)) * 2u);
    }
}

__device__ __forceinline__ bf16x8 nt2_frag(const char* sT, int r, int c) {
    return *(const bf16x8*)(sT + r * 128 + (nt2_swz(r, c) << 4));
}

template <int MODE, int ACT>
__global__ __launch_bounds__(NT2, 2) void gemm_nt2_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         int M, int N, int K, int lda, int ldb, GemmEpilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = (N + BN2 - 1) / BN2;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * BM2, n0 = (bid % tiles_n) * BN2;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)M * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)N * (uint32_t)ldb * 2u);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    const int nk = K / BK2;
    auto stage = [&](int kt, int buf) {
        char* s = smem + buf * STAGE2;
        nt2_stage<4>(rsA, s, m0, lda, kt * BK2, wid, lane);
        nt2_stage<2>(rsB, s + A2_BYTES, n0, ldb, kt * BK2, wid, lane);
    };
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int cur = 0, nxt2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the newest stage (6 DMA per wave) is still outstanding
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everyone's part of tile kt landed; everyone is done with tile kt-1
        if (kt + 2 < nk) stage(kt + 2, nxt2);
        const char* sA = smem + cur * STAGE2;
        const char* sB = sA + A2_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = kk * 2 + (lane >> 5);
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = nt2_frag(sA, wr * 64 + i * 32 + (lane & 31), c);
                b[i] = nt2_frag(sB, wc * 64 + i * 32 + (lane & 31), c);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(b[j], a[i], acc[i][j]);
        }
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }

    // (B-frag, A-frag) operand order: lane owns row m = lane&31 and, for g = 0..3, the 4
    // consecutive columns n = 8g + 4*(lane>>5) + 0..3 of each 32x32 tile (regs 4g..4g+3).
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wr * 64 + i * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= N) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                v *= ep.alpha;
                if (ep.bias) v += *(const f32x4*)(ep.bias + n);
                const size_t o = (size_t)m * ep.ldc + n;
                if constexpr (MODE == GEMM_STORE) {
                    if constexpr (ACT != ACT_NONE) {
                        if (ep.aux_out) {
                            bf16x4 p = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                            *(bf16x4*)(ep.aux_out + o) = p;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ACT == ACT_QUICKGELU ? quick_gelu_f(v[e]) : gelu_erf_f(v[e]);
                    }
                } else if constexpr (MODE == GEMM_RESID) {
                    v += *(const f32x4*)(ep.resid + (size_t)m * ep.ldr + n);
                } else if constexpr (MODE == GEMM_DACT) {
                    const bf16x4 p = *(const bf16x4*)(ep.aux_in + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = bf2f(p[e]);
                        v[e] *= ACT == ACT_QUICKGELU ? quick_gelu_grad_f(x) : gelu_erf_grad_f(x);
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

template <int MODE, int ACT>
static int launch_nt2(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, const GemmEpilogue& ep,
                      int tiles, hipStream_t st) {
    auto kern = gemm_nt2_kernel<MODE, ACT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NT2), LDS2, st, A, B, M, N, K, lda, ldb, ep);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int gemm_nt2(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode, const GemmEpilogue& ep,
             hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return SPN_ERR_ARG;
    if (K % BK2 || N % 4 || lda % 8 || ldb % 8 || ep.ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)M * lda * 2 >= (1ull << 32) || (uint64_t)N * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (!ep.out_f32 && !ep.out_bf16) return SPN_ERR_ARG;
    const int tiles = ((M + BM2 - 1) / BM2) * ((N + BN2 - 1) / BN2);
    ProfScope prof(PK_GEMM_NT, 2.0 * M * N * K, st);
    if (mode == GEMM_STORE) {
        if (ep.act == ACT_NONE) return launch_nt2<GEMM_STORE, ACT_NONE>(A, B, M, N, K, lda, ldb, ep, tiles, st);
        if (ep.act == ACT_QUICKGELU) return launch_nt2<GEMM_STORE, ACT_QUICKGELU>(A, B, M, N, K, lda, ldb, ep, tiles, st);
        if (ep.act == ACT_GELU_ERF) return launch_nt2<GEMM_STORE, ACT_GELU_ERF>(A, B, M, N, K, lda, ldb, ep, tiles, st);
        return SPN_ERR_ARG;
    }
    if (mode == GEMM_RESID) {
        if (!ep.resid || !ep.out_f32) return SPN_ERR_ARG;
        return launch_nt2<GEMM_RESID, ACT_NONE>(A, B, M, N, K, lda, ldb, ep, tiles, st);
    }
    if (mode == GEMM_DACT) {
        if (!ep.aux_in) return SPN_ERR_ARG;
        if (ep.act == ACT_QUICKGELU) return launch_nt2<GEMM_DACT, ACT_QUICKGELU>(A, B, M, N, K, lda, ldb, ep, tiles, st);
        if (ep.act == ACT_GELU_ERF) return launch_nt2<GEMM_DACT, ACT_GELU_ERF>(A, B, M, N, K, lda, ldb, ep, tiles, st);
    }
    return SPN_ERR_ARG;
}

// ----------------------------------------------------------------------------------- TN
// LDS image of a [64 k][COLS] bf16 tile (COLS = 256 for A, 128 for B): row k at byte k*2*COLS;
// the 32-byte chunk holding logical columns 16c..16c+15 sits at chunk position c ^ ((k&3)<<1):
// the two 16-lane groups of a half-wave read chunks c, c+1 of rows k0..k0+3 -> 8 distinct
// 32-byte bank groups.
template <int COLS, int PER_WAVE>
__device__ __forceinline__ void tn2_stage(__amdgpu_buffer_rsrc_t rs, char* sT, int kbase, int ld, int col0, int wid,
                                          int lane) {
    constexpr int ROWB = COLS * 2;
    constexpr int ROWS_PER_INSTR = 1024 / ROWB;        // 2 (A) or 4 (B)
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

__global__ __launch_bounds__(NT2, 2) void gemm_tn2_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                         int Kr, int N1, int N2, int lda, int ldb,
                                                         float* __restrict__ C, int ldc, size_t split_stride,
                                                         int k_chunk, float* __restrict__ colsum_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = (N2 + BN2 - 1) / BN2;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * BM2, n0 = (bid % tiles_n) * BN2;
    const int kb = blockIdx.y * k_chunk;
    const int ke = min(Kr, kb + k_chunk);
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)Kr * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)Kr * (uint32_t)ldb * 2u);
    const bool do_colsum = colsum_out != nullptr && n0 == 0 && wc == 0;

    f32x16 acc[2][2], accs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        accs[i] = zero16();
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    const int nk = (ke - kb + BK2 - 1) / BK2;
    auto stage = [&](int kt, int buf) {
        char* s = smem + buf * STAGE2;
        tn2_stage<256, 4>(rsA, s, kb + kt * BK2, lda, m0, wid, lane);
        tn2_stage<128, 2>(rsB, s + A2_BYTES, kb + kt * BK2, ldb, n0, wid, lane);
    };
    if (nk > 0) stage(0, 0);
    if (nk > 1) stage(1, 1);
    int cur = 0, nxt2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(kt + 2, nxt2);
        const char* sA = smem + cur * STAGE2;
        const char* sB = sA + A2_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = tn2_frag<256>(sA, wr * 64 + i * 32, kk, lane);
                b[i] = tn2_frag<128>(sB, wc * 64 + i * 32, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(b[j], a[i], acc[i][j]);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 2; ++i) accs[i] = mfma32(ones, a[i], accs[i]);
            }
        }
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    float* Cz = C + (size_t)blockIdx.y * split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wr * 64 + i * 32 + (lane & 31);
        if (m >= N1) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                if (n >= N2) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *(f32x4*)(Cz + (size_t)m * ldc + n) = v;
            }
        // every row of the ones-product equals the column sum; lanes 0..31 hold columns m
        if (do_colsum && lane < 32) colsum_out[(size_t)blockIdx.y * N1 + m] = accs[i][0];
    }
}

__global__ void splitk_reduce2_kernel(const float* __restrict__ ws, int splits, int rows, int cols,
                                      float* __restrict__ out, int ldo, float alpha, int accumulate) {
    const int c4 = cols >> 2;
    const size_t total = (size_t)rows * c4;
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

static int tn2_splits(int Kr, int N1, int N2) {
    const int tiles = ((N1 + BM2 - 1) / BM2) * ((N2 + BN2 - 1) / BN2);
    const int ktiles = (Kr + BK2 - 1) / BK2;
    int s = 256 / tiles;
    if (s < 1) s = 1;
    const int max_s = (ktiles + 5) / 6;   // >= ~6 k-tiles per split keeps the 3-stage pipe busy
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

size_t gemm_tn2_workspace_bytes(int Kr, int N1, int N2) {
    const int s = tn2_splits(Kr, N1, N2);
    return ((size_t)s * N1 * N2 + (size_t)s * N1) * sizeof(float);
}

// colsum_out (optional, [N1]): column sums of A over all Kr rows (overwritten)
int gemm_tn2(const bf16_t* A, const bf16_t* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
             float alpha, int accumulate, float* colsum_out, float* ws, size_t ws_bytes, hipStream_t st) {
    if (Kr <= 0 || N1 <= 0 || N2 <= 0) return SPN_ERR_ARG;
    if (N1 % 8 || N2 % 8 || lda % 8 || ldb % 8 || ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)Kr * lda * 2 >= (1ull << 32) || (uint64_t)Kr * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (ws_bytes < gemm_tn2_workspace_bytes(Kr, N1, N2)) return SPN_ERR_WORKSPACE;
    const int tiles = ((N1 + BM2 - 1) / BM2) * ((N2 + BN2 - 1) / BN2);
    int splits = tn2_splits(Kr, N1, N2);
    const int ktiles = (Kr + BK2 - 1) / BK2;
    const int k_chunk = ((ktiles + splits - 1) / splits) * BK2;
    splits = (Kr + k_chunk - 1) / k_chunk;
    float* cs_ws = colsum_out ? ws + (size_t)splits * N1 * N2 : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    {
        ProfScope prof(PK_GEMM_TN, 2.0 * Kr * N1 * N2, st);
        hipLaunchKernelGGL(gemm_tn2_kernel, dim3(tiles, splits), dim3(NT2), LDS2, st, A, B, Kr, N1, N2, lda, ldb, ws, N2,
                           (size_t)N1 * N2, k_chunk, cs_ws);
    }
    SPN_CHECK_LAUNCH();
    const size_t total = (size_t)N1 * (N2 / 4);
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce2_kernel, dim3(blocks), dim3(256), 0, st, ws, splits, N1, N2, C, ldc, alpha, accumulate);
    SPN_CHECK_LAUNCH();
    if (colsum_out) return fold_rows(cs_ws, (size_t)N1, splits, (size_t)N1, colsum_out, 1.0f, 0, st);
    return SPN_OK;
}

}  // namespace spn
