// bf16 MFMA GEMMs for the text / vision / fusion towers (gfx950).
//
//   gemm_nt : C[M,N]  = epilogue( A[M,K] . B[N,K]^T )      both operands K-contiguous
//   gemm_tn : C[N1,N2] = A[Kr,N1]^T . B[Kr,N2]              both operands reduction-major
//                                                           (weight gradients: dW = dY^T X)
//
// 128x128 output tile, 64-deep K step, 256 threads = 4 waves (2x2), each wave 64x64 =
// 4x4 v_mfma_f32_16x16x32_bf16 tiles, fp32 accumulation.  Operand tiles are DMA'd straight
// into LDS with buffer_load ... lds (16 B / lane, out-of-range rows read as zero) into an
// XOR-swizzled image (swizzle applied to the per-lane SOURCE address, cdna guide rule 21)
// so that the ds_read_b128 / ds_read_b64_tr_b16 fragment reads are bank-conflict free.
// Two LDS stages: tile t+1 streams in while tile t is multiplied.
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "gemm_v1_tiles.h"

namespace spn {

// MI = 16-row MFMA tiles per wave along M: 4 -> the 128x128 tile (2 blocks per CU); 2 -> 64x128 (24 KiB per stage,
// 3 blocks per CU) for products with fewer 128x128 tiles than CUs (the BERT-side GEMMs of the fusion encoder: 4 096 rows).
// Round 3: the 64-row variant keeps THREE stages (72 KB, two workgroups per CU) with counted waits and a raw barrier: with
// two stages and `wait all; barrier; request next; compute` every k tile of a workgroup waited out the L2 latency of the
// tile it had just requested, hidden only by the two other workgroups of the CU - and a product that needs this kernel has
// about two workgroups per CU to begin with (the packed text tower: 492 workgroups).
template <int MI>
static constexpr int nt_v1_stages() { return MI == 2 ? 3 : 2; }

template <int MODE, int ACT, int MI = 4>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(const bf16_t* __restrict__ A,
                                                              const bf16_t* __restrict__ B, int M, int N, int K,
                                                              int lda, int ldb, GemmEpilogue ep) {
    constexpr int BM = 32 * MI;
    constexpr int A_BYTES = BM * 64 * 2, STAGE = A_BYTES + TILE_BYTES;
    constexpr int NS = nt_v1_stages<MI>();
    extern __shared__ __attribute__((aligned(16))) char smem[];          // NS x STAGE
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)M * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)N * (uint32_t)ldb * 2u);

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    auto stage = [&](int kt, int buf) {
        char* dst = smem + buf * STAGE;
        nt_stage<MI>(rsA, dst, m0, lda, kt * BK, wid, lane);
        nt_stage<4>(rsB, dst + A_BYTES, n0, ldb, kt * BK, wid, lane);
    };
    stage(0, 0);
    if constexpr (NS == 3) {
        if (nk > 1) stage(1, 1);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = NS == 3 ? kt % 3 : (kt & 1);
        if constexpr (NS == 3) {
            if (kt + 1 < nk) wait_vmcnt<MI + 4>();      // tile kt is in; tile kt + 1 (MI + 4 DMA per wave) may still be in flight
            else wait_vmcnt<0>();
            lds_barrier();                               // raw barrier: __syncthreads() would drain the tile in flight (common.h)
            if (kt + 2 < nk) stage(kt + 2, (kt + 2) % 3);   // = the buffer of tile kt - 1, read by everyone before this barrier
        } else {
            wait_vm0();
            __syncthreads();   // tile kt landed for every wave; everyone finished reading buf^1
            if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        }
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[MI], b[4];
            const int c = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = nt_frag(sA, wr * (16 * MI) + i * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = nt_frag(sB, wc * 64 + i * 16 + (lane & 15), c);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
        }
    }

    // Epilogue.  With the (B-fragment, A-fragment) operand order each lane owns, per 16x16
    // tile, row m = lane&15 and the 4 consecutive columns n = (lane>>4)*4 .. +3.
    // Every operand the epilogue reads (bias, residual, act'(pre)) is fetched BEFORE the first store: the residual may alias the
    // output (in-place add), so the compiler keeps a load behind the store in front of it - 2 MI dependent L2 round trips per
    // tile, 4-6 us of a 25 us launch on the BERT-side products of the fusion encoder.
    f32x4 bias4[4];
    [[maybe_unused]] f32x4 res4[MODE == GEMM_RESID ? MI : 1][4];
    [[maybe_unused]] bf16x4 aux4[MODE == GEMM_DACT ? MI : 1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
        bias4[j] = (ep.bias && n < N) ? *(const f32x4*)(ep.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wr * (16 * MI) + i * 16 + (lane & 15);
            const bool ok = m < M && n < N;
            if constexpr (MODE == GEMM_RESID)
                res4[i][j] = ok ? *(const f32x4*)(ep.resid + (size_t)m * ep.ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE == GEMM_DACT) {
                const bf16x4 z = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                aux4[i][j] = ok ? *(const bf16x4*)(ep.aux_in + (size_t)m * ep.ldc + n) : z;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wr * (16 * MI) + i * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            f32x4 v = acc[i][j] * ep.alpha;
            v += bias4[j];
            const size_t o = (size_t)m * ep.ldc + n;
            if constexpr (MODE == GEMM_STORE) {
                if constexpr (ACT != ACT_NONE) {
                    f32x4 a = v;
                    if (ep.aux_out && ep.aux_grad) {           // wave-uniform: keep act'(pre) instead of pre
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
                v += res4[i][j];
            } else if constexpr (MODE == GEMM_DACT) {
                const bf16x4 p = aux4[i][j];
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

bool gemm_use_v1() {
    return gemm_cfg() == 0;
}

// SPN_GEMM_CFG: 0 = v1 kernels of this file (128x128, 2 blocks/CU); 1/2/3 = gemm2.hip tile configs (default 3)
int gemm_cfg() {
    static const int cfg = [] {
        const char* e = spn_env("SPN_GEMM_CFG");
        return e ? atoi(e) : 3;   // measured best on the tower shapes: 256x256 tile, staged epilogue
    }();
    return cfg;
}

template <int MODE, int ACT, int MI>
static int launch_nt_v1(int tiles, hipStream_t st, const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb,
                        const GemmEpilogue& ep) {
    constexpr int LDS = nt_v1_stages<MI>() * (32 * MI * 64 * 2 + TILE_BYTES);
    auto kern = gemm_nt_kernel<MODE, ACT, MI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), LDS, st, A, B, M, N, K, lda, ldb, ep);
    return SPN_OK;
}

int gemm_nt(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode,
            const GemmEpilogue& ep, hipStream_t st) {
    // 256x256 tiles need roughly half of the 256 CUs worth of tiles to pay off (measured break-even between 63 and 150
    // tiles); small products (the packed text tower:
    // M ~ 5 k rows, N = 768 -> 63 tiles) run on the 128x128 kernel below, which brings 4x the workgroups.
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long t128 = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const bool small = gemm_cfg() == 3 && t256 < 112 && t128 >= 2 * t256 && mode != GEMM_BANKSTATS;
#ifdef SPN_EXPERIMENTS
    {   // SPN_NT_MID_ALL=1 with SPN_NT_MID=v: EVERY product on the mid-size tile variant v (A/B against the 256 x 256 kernel)
        static const int mid_all = [] { const char* e = spn_env("SPN_NT_MID_ALL"); return e ? atoi(e) : 0; }();
        static const int mid_v = [] { const char* e = spn_env("SPN_NT_MID"); return e ? atoi(e) : 0; }();
        if (mid_all && mid_v > 0 && mode != GEMM_BANKSTATS && !gemm_use_v1()) return gemm_nt2_mid(A, B, M, N, K, lda, ldb, mode, ep, st, mid_v);
    }
#endif
    if (!gemm_use_v1() && !small) return gemm_nt2(A, B, M, N, K, lda, ldb, mode, ep, st);
#ifdef SPN_EXPERIMENTS
    // SPN_NT_MID=1..4 (experiments build; measured slower, gemm2.hip: gemm_nt2_mid): the small products on a mid-size tile of the
    // second-generation kernel
    static const int nt_mid = [] { const char* e = spn_env("SPN_NT_MID"); return e ? atoi(e) : 0; }();
    if (small && nt_mid > 0 && !gemm_use_v1()) return gemm_nt2_mid(A, B, M, N, K, lda, ldb, mode, ep, st, nt_mid);
#endif
    if (M <= 0 || N <= 0 || K <= 0) return SPN_ERR_ARG;
    if (K % BK || N % 4 || lda % 8 || ldb % 8 || ep.ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)M * lda * 2 >= (1ull << 32) || (uint64_t)N * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    if (!ep.out_f32 && !ep.out_bf16) return SPN_ERR_ARG;
    // fewer 128x128 tiles than CUs: 64-row tiles, three workgroups per CU (SPN_NT_SMALL_MI=4 keeps the 128-row tile)
    static const int small_mi = [] { const char* e = spn_env("SPN_NT_SMALL_MI"); return e ? atoi(e) : 2; }();
    const bool half = small_mi == 2 && t128 < 256;
    const int tiles = ((M + (half ? 64 : BM) - 1) / (half ? 64 : BM)) * ((N + BN - 1) / BN);
    ProfScope prof(PK_GEMM_NT, 2.0 * M * N * K, st);
#define SPN_LAUNCH_NT(MODE_, ACT_)                                                                                       \
    do {                                                                                                                 \
        const int rc_ = half ? launch_nt_v1<MODE_, ACT_, 2>(tiles, st, A, B, M, N, K, lda, ldb, ep)                      \
                             : launch_nt_v1<MODE_, ACT_, 4>(tiles, st, A, B, M, N, K, lda, ldb, ep);                     \
        if (rc_) return rc_;                                                                                             \
    } while (0)
    if (mode == GEMM_STORE) {
        if (ep.act == ACT_NONE) SPN_LAUNCH_NT(GEMM_STORE, ACT_NONE);
        else if (ep.act == ACT_QUICKGELU) SPN_LAUNCH_NT(GEMM_STORE, ACT_QUICKGELU);
        else if (ep.act == ACT_GELU_ERF) SPN_LAUNCH_NT(GEMM_STORE, ACT_GELU_ERF);
        else return SPN_ERR_ARG;
    } else if (mode == GEMM_RESID) {
        if (!ep.resid || !ep.out_f32) return SPN_ERR_ARG;
        SPN_LAUNCH_NT(GEMM_RESID, ACT_NONE);
    } else if (mode == GEMM_DACT) {
        if (!ep.aux_in) return SPN_ERR_ARG;
        if (ep.act == ACT_QUICKGELU) SPN_LAUNCH_NT(GEMM_DACT, ACT_QUICKGELU);
        else if (ep.act == ACT_GELU_ERF) SPN_LAUNCH_NT(GEMM_DACT, ACT_GELU_ERF);
        else return SPN_ERR_ARG;
    } else {
        return SPN_ERR_ARG;
    }
#undef SPN_LAUNCH_NT
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ---------------------------------------------------------------------------------------
// TN: C[N1,N2] (fp32) = sum_k A[k][n1] * B[k][n2], optional split over k (grid.y).
// LDS image of a [64 k][128 n] bf16 tile: row k at byte k*256; the 32-byte chunk holding
// logical columns 16*c..16*c+15 sits at position c ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2, so
// the 8 rows touched by one 32-lane half of a ds_read_b64_tr_b16 hit 8 distinct bank groups.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A,
                                                              const bf16_t* __restrict__ B, int Kr, int N1, int N2,
                                                              int lda, int ldb, float* __restrict__ C, int ldc,
                                                              size_t split_stride, int k_chunk,
                                                              float* __restrict__ colsum_out) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_n = (N2 + BN - 1) / BN;
    // XCD-aware order over the WHOLE (tile, split) grid: the tiles of one k split are contiguous logical ids, so they run on
    // one XCD and share their operand panels through its L2.  With the remap over the tiles only, the hardware's linear
    // order (x + tiles * y) scattered the 12 tiles of a split of the bank's dq GEMM over all eight XCDs: PMC FETCH_SIZE
    // 200 MB for 81 MB of operands.
    const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
    const int logical = xcd_remap(lin, (int)(gridDim.x * gridDim.y));
    const int bid = logical % (int)gridDim.x, split_z = logical / (int)gridDim.x;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int kb = split_z * k_chunk;
    const int ke = min(Kr, kb + k_chunk);
    // bias gradient for free: an all-ones B operand makes every output row the column sum of A
    const bool do_colsum = colsum_out != nullptr && n0 == 0 && wc == 0;
    f32x4 accs[4];
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (uint32_t)Kr * (uint32_t)lda * 2u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (uint32_t)Kr * (uint32_t)ldb * 2u);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (ke - kb + BK - 1) / BK;
    if (nk > 0) {
        tn_stage(rsA, smem, kb, lda, m0, wid, lane);
        tn_stage(rsB, smem + TILE_BYTES, kb, ldb, n0, wid, lane);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        wait_vm0();
        __syncthreads();
        if (kt + 1 < nk) {
            char* nxt = smem + (buf ^ 1) * 2 * TILE_BYTES;
            tn_stage(rsA, nxt, kb + (kt + 1) * BK, lda, m0, wid, lane);
            tn_stage(rsB, nxt + TILE_BYTES, kb + (kt + 1) * BK, ldb, n0, wid, lane);
        }
        const char* sA = smem + buf * 2 * TILE_BYTES;
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            TnFrag fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = tn_frag(sA, wr * 64 + i * 16, ks, lane);
                fb[i] = tn_frag(sB, wc * 64 + i * 16, ks, lane);
            }
            wait_lgkm<0>();
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = tn_tie(fa[i]);
                b[i] = tn_tie(fb[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accs[i] = mfma16(ones, a[i], accs[i]);
            }
        }
    }
    float* Cz = C + (size_t)split_z * split_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= N1) continue;
        if (do_colsum && lane < 16) colsum_out[(size_t)split_z * N1 + m] = accs[i][0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= N2) continue;
            *(f32x4*)(Cz + (size_t)m * ldc + n) = acc[i][j];
        }
    }
}

// out[r*ldo + c] = (accumulate ? out : 0) + alpha * sum_z ws[z][r][c]
// 64 column quads x 4 split groups per workgroup: a lane's loads are independent (4-way unrolled), the groups meet in LDS -
// one lane walking all splits of its quad alone is a chain of `splits` dependent-latency loads (13 us for 42 x 786 KB)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int rows, int cols,
                                                            float* __restrict__ out, int ldo, float alpha, int accumulate) {
    __shared__ f32x4 part[4][64];
    const int c4 = cols >> 2;
    const size_t total = (size_t)rows * c4;
    const int e = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const size_t slab = (size_t)rows * cols;
    for (size_t i0 = (size_t)blockIdx.x * 64; i0 < total; i0 += (size_t)gridDim.x * 64) {
        const size_t i = i0 + e;
        const bool in = i < total;
        const int r = in ? (int)(i / c4) : 0, c = in ? (int)(i % c4) * 4 : 0;
        const float* p = ws + (size_t)r * cols + c;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (in) {
            int z = zg;
            for (; z + 12 < splits; z += 16) {
                const f32x4 a = *(const f32x4*)(p + (size_t)z * slab), b = *(const f32x4*)(p + (size_t)(z + 4) * slab);
                const f32x4 d = *(const f32x4*)(p + (size_t)(z + 8) * slab), f = *(const f32x4*)(p + (size_t)(z + 12) * slab);
                s += (a + b) + (d + f);
            }
            for (; z < splits; z += 4) s += *(const f32x4*)(p + (size_t)z * slab);
        }
        part[zg][e] = s;
        __syncthreads();
        if (zg == 0 && in) {
            s = ((part[0][e] + part[1][e]) + (part[2][e] + part[3][e])) * alpha;
            float* o = out + (size_t)r * ldo + c;
            if (accumulate) s += *(const f32x4*)o;
            *(f32x4*)o = s;
        }
        __syncthreads();
    }
}

int gemm_tn_splits(int Kr, int N1, int N2) {
    const int tiles = ((N1 + BM - 1) / BM) * ((N2 + BN - 1) / BN);
    const int ktiles = (Kr + BK - 1) / BK;
    int s = 512 / tiles;
    if (s < 1) s = 1;
    const int max_s = (ktiles + 3) / 4;   // at least ~4 k-tiles per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

size_t gemm_tn_workspace_bytes(int Kr, int N1, int N2) {
    const int s = gemm_tn_splits(Kr, N1, N2);
    const size_t v1 = ((size_t)s * N1 * N2 + (size_t)s * N1) * sizeof(float);
    const size_t v2 = gemm_tn2_workspace_bytes(Kr, N1, N2);
    return v1 > v2 ? v1 : v2;
}

int gemm_tn(const bf16_t* A, const bf16_t* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
            float alpha, int accumulate, float* colsum_out, float* ws, size_t ws_bytes, hipStream_t st) {
    // small outputs (e.g. 768x768) give the 256x256 tiling too few workgroups: v1 (128x128) is faster there
    static const size_t tn2_min = [] {
        const char* e = spn_env("SPN_TN2_MIN");           // experiment knob: smallest N1*N2 routed to the 256x256 kernel
        return e ? (size_t)atoll(e) : (size_t)768 * 2304;
    }();
    if (!gemm_use_v1() && (size_t)N1 * N2 >= tn2_min)
        return gemm_tn2(A, B, Kr, N1, N2, lda, ldb, C, ldc, alpha, accumulate, colsum_out, ws, ws_bytes, st);
    if (Kr <= 0 || N1 <= 0 || N2 <= 0) return SPN_ERR_ARG;
    if (N1 % 8 || N2 % 8 || lda % 8 || ldb % 8 || ldc % 4) return SPN_ERR_SHAPE;
    if ((uint64_t)Kr * lda * 2 >= (1ull << 32) || (uint64_t)Kr * ldb * 2 >= (1ull << 32)) return SPN_ERR_SHAPE;
    const int tiles = ((N1 + BM - 1) / BM) * ((N2 + BN - 1) / BN);
    int splits = gemm_tn_splits(Kr, N1, N2);
    if (ws_bytes < ((size_t)splits * N1 * N2 + (size_t)splits * N1) * sizeof(float)) return SPN_ERR_WORKSPACE;
    const int ktiles = (Kr + BK - 1) / BK;
    const int k_chunk = ((ktiles + splits - 1) / splits) * BK;
    splits = (Kr + k_chunk - 1) / k_chunk;
    float* cs_ws = colsum_out ? ws + (size_t)splits * N1 * N2 : nullptr;
    if (splits == 1 && !accumulate && alpha == 1.0f) {   // single split: straight into C
        ProfScope prof(PK_GEMM_TN, 2.0 * Kr * N1 * N2, st);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, 1), dim3(NTHREADS), 0, st, A, B, Kr, N1, N2, lda, ldb, C, ldc,
                           (size_t)0, k_chunk, colsum_out);
        SPN_CHECK_LAUNCH();
        return SPN_OK;
    }
    {
        ProfScope prof(PK_GEMM_TN, 2.0 * Kr * N1 * N2, st);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, splits), dim3(NTHREADS), 0, st, A, B, Kr, N1, N2, lda, ldb, ws,
                           N2, (size_t)N1 * N2, k_chunk, cs_ws);
    }
    SPN_CHECK_LAUNCH();
    const size_t total = (size_t)N1 * (N2 / 4);
    const int blocks = (int)((total + 63) / 64 > 4096 ? 4096 : (total + 63) / 64);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, splits, N1, N2, C, ldc, alpha,
                       accumulate);
    SPN_CHECK_LAUNCH();
    if (colsum_out) return fold_rows(cs_ws, (size_t)N1, splits, (size_t)N1, colsum_out, 1.0f, 0, st);
    return SPN_OK;
}

}  // namespace spn
