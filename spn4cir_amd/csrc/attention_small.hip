// Whole-head attention kernels for short self-attention (Lq == Lk = L <= 128): the CLIP text tower
// (L = 77, causal) and BERT self-attention.  One workgroup per (batch, head), one wave per 16-row
// tile (5 waves at L = 77): Q, K, V (and dO) are loaded into LDS ONCE, the full score rows live in
// registers (no online softmax), and every transposed operand comes from ds_read_b64_tr_b16 on the
// row-major tiles - no transposed LDS copies, no second pass over HBM.
//
// Backward recomputes the softmax from S (it has whole rows) and uses delta_q = sum_k P dP, so it
// needs neither the forward's LSE nor O:
//   phase 1 (wave = query tile): S, dP -> P, dS ; dQ = dS K         ; P, dS also go to LDS
//   phase 2 (wave = key tile)  : dV = P^T dO ; dK = dS^T Q          (transpose reads of P, dS, dO, Q)
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

#ifndef SPN_ATTN_ABL
#define SPN_ATTN_ABL 0   // ablation builds of attention_small_bwd_kernel: 1 = no phase 1, 2 = no phase 2 MFMAs, 4 = no output stores
#endif
static constexpr int SHD = 64;       // head dim
static constexpr int SLD = 72;       // LDS row stride (elements) of the [row][64] tiles: 144 B

__device__ __forceinline__ f32x4 mfma16s(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 zero8s() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.0f;
    return z;
}

// rows [0, ROWS) x 64 columns of a [L, ld] matrix (base offset to batch/head) -> s[r][c], rows >= L zero
template <int ROWS>
__device__ __forceinline__ void load_rows(const bf16_t* __restrict__ base, int ld, int L, int tid, int nthreads,
                                          bf16_t* s) {
    for (int i = tid; i < ROWS * 8; i += nthreads) {
        const int r = i >> 3, ch = i & 7;
        bf16x8 v = zero8s();
        if (r < L) v = *(const bf16x8*)(base + (size_t)r * ld + ch * 8);
        *(bf16x8*)(s + r * SLD + ch * 8) = v;
    }
}

// The same for NSRC matrices at once: every global load of the block is in flight before the first LDS store
// (the one-matrix loop above serialises load -> wait -> store per 16 B and per matrix).
template <int ROWS, int NTHREADS, int NSRC>
__device__ __forceinline__ void load_rows_multi(const bf16_t* const (&base)[NSRC], const int (&ld)[NSRC], int L, int tid,
                                                bf16_t* const (&dst)[NSRC]) {
    constexpr int ITER = (ROWS * 8 + NTHREADS - 1) / NTHREADS;
    bf16x8 v[NSRC][ITER];
#pragma unroll
    for (int k = 0; k < NSRC; ++k)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTHREADS, r = i >> 3, ch = i & 7;
            v[k][it] = zero8s();
            if (i < ROWS * 8 && r < L) v[k][it] = *(const bf16x8*)(base[k] + (size_t)r * ld[k] + ch * 8);
        }
#pragma unroll
    for (int k = 0; k < NSRC; ++k)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTHREADS, r = i >> 3, ch = i & 7;
            if (i < ROWS * 8) *(bf16x8*)(dst[k] + r * SLD + ch * 8) = v[k][it];
        }
}

// fragment: "row" index (lane&15) walks LDS rows, 8 k-values contiguous along the row
__device__ __forceinline__ bf16x8 rfrag(const bf16_t* s, int ld, int row_base, int kcol, int lane) {
    return *(const bf16x8*)(s + (row_base + (lane & 15)) * ld + kcol + (lane >> 4) * 8);
}

// transposed fragment: index (lane&15) walks the contiguous dim (columns cb..cb+15), the 8 k-values
// walk LDS rows kbase + (lane>>4)*8 + 0..7
// Rows >= zrow are redirected to row zrow, which holds zeros (keeps the tiles at NT*16+1 rows instead
// of padding them to a multiple of 32).
__device__ __forceinline__ bf16x8 tfrag(const bf16_t* s, int ld, int kbase, int cb, int lane, int zrow) {
    union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int row = kbase + (lane >> 4) * 8 + h * 4 + ((lane & 15) >> 2);
        row = row < zrow ? row : zrow;
        u.h[h] = lds_tr16_b64(s + row * ld + cb + (lane & 3) * 4);
    }
    return u.v;
}

__device__ __forceinline__ float qg_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float qg_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Output path of the whole-head kernels.  A wave holds a [16 rows][64] tile as four MFMA accumulators (row = lane & 15,
// columns 16 d + 4 (lane >> 4) + r): written from there a store instruction covers 64 pieces of 8 B in 16 rows.  Staged
// through LDS rows `st` (stride `ld` elements, a region only this wave touches; 144 / 208-byte strides put the 8-byte
// writes of 16 rows in distinct banks) and read back 16 B per lane, a store covers 8 whole 128-byte rows instead
// (cross-attention backward 153 -> 112 us).  dst_row(i) = global address of tile row i, or nullptr to skip it.
template <class F>
__device__ __forceinline__ void store_rows_staged(bf16_t* st, int ld, const f32x4 (&acc)[4], float scale, int lane,
                                                  F dst_row) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        bf16x4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = f2bf(acc[d][r] * scale);
        *(bf16x4*)(st + (lane & 15) * ld + d * 16 + (lane >> 4) * 4) = ob;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3);
        const bf16x8 x = *(const bf16x8*)(st + row * ld + (lane & 7) * 8);
        bf16_t* p = dst_row(row);
        if (p && !((SPN_ATTN_ABL & 4) && x[0] != (bf16_t)123.0f)) *(bf16x8*)(p + (lane & 7) * 8) = x;
    }
}

// NT = number of 16-row tiles (= waves); keys are padded to KP = 32*ceil(NT/2) rows of zeros
// BIAS (compile time): an additive per-key bias (BERT's padding mask) is present.  As a run-time test per logit the
// compiler turned it into one branch + global load + vmcnt(0) per element, which cut the softmax phase into ~20 basic
// blocks per wave (backward, L = 77: phase 1 took 31 of 74 us).
template <int NT, bool BIAS>
__global__ __launch_bounds__(NT * 64) void attention_small_fwd_kernel(AttnArgs a) {
    constexpr int KS = (NT + 1) / 2, KP = KS * 32;     // 32-deep key steps, padded key rows
    constexpr int PLD = KP + 8;                         // P row stride (elements)
    // LDS: [V tile][K tile | tail]; the per-wave P rows are written over the K tile once every wave has its logits
    // (one extra barrier), which brings a block from 44 KB to 30 KB: 5 instead of 3 resident blocks per CU - the
    // kernel is bound by loads in flight, not by anything it computes (36.6 -> 30.8 us at B = 256, H = 12, L = 77)
    constexpr int PK = NT * 16 * PLD > KP * SLD ? NT * 16 * PLD : KP * SLD;
    __shared__ __attribute__((aligned(16))) bf16_t smem[KP * SLD + PK];
    bf16_t* Vs = smem;
    bf16_t* Ks = smem + KP * SLD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    bf16_t* Ps = Ks + w * 16 * PLD;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int L = a.cu ? a.cu[b + 1] - a.cu[b] : a.Lq;                 // this sequence's length
    const size_t row0 = a.cu ? (size_t)a.cu[b] : (size_t)b * a.Lq;     // its first row
    const bf16_t* qb = a.q + row0 * a.ldq + h * SHD;
    const bf16_t* kb = a.k + row0 * a.ldk + h * SHD;
    const bf16_t* vb = a.v + row0 * a.ldv + h * SHD;
    [[maybe_unused]] const float* kbias = BIAS ? a.key_bias + (size_t)b * a.Lq : nullptr;
    {
        const bf16_t* const bases[2] = {kb, vb};
        const int lds_[2] = {a.ldk, a.ldv};
        bf16_t* const dsts[2] = {Ks, Vs};
        load_rows_multi<KP, NT * 64, 2>(bases, lds_, L, tid, dsts);
    }
    const int q0 = w * 16, qrow = q0 + (lane & 15);
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = zero8s();
        if (qrow < L) qf[ks] = *(const bf16x8*)(qb + (size_t)qrow * a.ldq + ks * 32 + (lane >> 4) * 8);
    }
    __syncthreads();
    const bool active = q0 < L;                         // inactive waves still meet the barrier below
    const int nkt = a.causal ? w + 1 : NT;              // key tiles this query tile can see
    f32x4 s[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        s[kt] = f32x4{0, 0, 0, 0};
        if (kt < nkt && active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[kt] = mfma16s(rfrag(Ks, SLD, kt * 16, ks * 32, lane), qf[ks], s[kt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + (lane >> 4) * 4 + r;
            float v = s[kt][r] * a.scale;
            if constexpr (BIAS) v += kbias[min(key, max(L - 1, 0))];     // clamped, unconditional load; masked below
            // '|' on purpose: the short-circuit form compiles to three exec-mask branches per logit
            const bool dead = (kt >= nkt) | (key >= L) | ((a.causal != 0) & (key > qrow));
            v = dead ? -INFINITY : v;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = qg_max(mx);
    __syncthreads();                                    // every wave is done with the K tile: P goes over it
    if (!active) return;
    const float m_use = mx == -INFINITY ? 0.f : mx;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2 * KS; ++kt) {
        bf16x4 pb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p = 0.f;
            if (kt < NT) p = __expf(s[kt < NT ? kt : 0][r] - m_use);
            sum += p;
            pb[r] = f2bf(p);
        }
        *(bf16x4*)(Ps + (lane & 15) * PLD + kt * 16 + (lane >> 4) * 4) = pb;
    }
    sum = qg_sum(sum);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0, 0, 0, 0};
    const int nks = a.causal ? (w + 2) / 2 : KS;        // key steps that hold visible keys
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks < nks) {
            const bf16x8 pf = rfrag(Ps, PLD, 0, ks * 32, lane);
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = mfma16s(tfrag(Vs, SLD, ks * 32, d * 16, lane, KP), pf, o[d]);
        }
    }
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    if constexpr (PLD >= 72) {                               // this wave's P rows are spent and wide enough for 64 columns
        store_rows_staged(Ps, PLD, o, inv, lane, [&](int row) -> bf16_t* {
            return q0 + row < L ? a.o + (row0 + q0 + row) * a.ldo + h * SHD : nullptr;
        });
    } else if (qrow < L) {
        bf16_t* orow = a.o + (row0 + qrow) * a.ldo + h * SHD;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = f2bf(o[d][r] * inv);
            *(bf16x4*)(orow + d * 16 + (lane >> 4) * 4) = ob;
        }
    }
    if (qrow >= L) return;
    if (a.lse && (lane >> 4) == 0)
        a.lse[((size_t)b * a.H + h) * a.Lq + qrow] = sum > 0.f ? mx + __logf(sum) : -INFINITY;
}

// Phase 1 of the whole-head backward for ONE wave: its 16 queries (rows q0..) against NKT key tiles (compile time).
// S = Q K^T, dP = dO V^T (fragments from the LDS tiles), softmax over the whole rows, delta = sum P dP, then P and dS =
// P (dP - delta) to their LDS matrices (bf16, zeros past NKT) and dQ^T = K^T dS^T.  Only the LAST visible tile can hold a
// masked logit (the causal diagonal or the end of the sequence): the others skip the mask arithmetic - the phase is bound by
// VALU issue, not by its MFMAs (23 of the kernel's 59 us before; key tiles past NKT used to pay masks, exponentials and
// conversions too).  Query rows >= L need no mask: their Q and dO rows are zero in LDS, so dS = 0 and P meets zero dO rows.
template <int NT, int NKT, bool BIAS>
__device__ __forceinline__ void attn_bwd_phase1(const bf16_t* Qs, const bf16_t* Ks, const bf16_t* Vs, const bf16_t* Os, bf16_t* Pm,
                                                bf16_t* Dm, int q0, int lane, int L, int causal, float scale,
                                                const float* kbias, f32x4 (&dq)[4]) {
    constexpr int KS = (NT + 1) / 2, KP = KS * 32, PLD = KP + 8, ZR = NT * 16;
    constexpr int NA = NKT > 0 ? NKT : 1;
    const int qrow = q0 + (lane & 15);
    f32x4 s[NA], dp[NA];
    float inv = 0.f, dl = 0.f;
    if constexpr (NKT > 0) {
        bf16x8 qf[2], dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = rfrag(Qs, SLD, q0, ks * 32, lane);
            dof[ks] = rfrag(Os, SLD, q0, ks * 32, lane);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            s[kt] = f32x4{0, 0, 0, 0};
            dp[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s[kt] = mfma16s(rfrag(Ks, SLD, kt * 16, ks * 32, lane), qf[ks], s[kt]);
                dp[kt] = mfma16s(rfrag(Vs, SLD, kt * 16, ks * 32, lane), dof[ks], dp[kt]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][r] * scale;
                if (BIAS || kt == NKT - 1) {                    // folded at compile time (the loop is unrolled)
                    const int key = kt * 16 + (lane >> 4) * 4 + r;
                    if constexpr (BIAS) v += kbias[min(key, max(L - 1, 0))];     // clamped, unconditional load; masked below
                    const bool dead = (key >= L) | ((causal != 0) & (key > qrow));   // no short circuit
                    v = dead ? -INFINITY : v;
                }
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = qg_max(mx);
        const float m_use = mx == -INFINITY ? 0.f : mx;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m_use);
                s[kt][r] = p;
                sum += p;
                dl += p * dp[kt][r];
            }
        sum = qg_sum(sum);
        dl = qg_sum(dl);
        inv = sum > 0.f ? 1.0f / sum : 0.f;
        dl *= inv;                                       // delta_q = sum_k P dP
    }
#pragma unroll
    for (int kt = 0; kt < 2 * KS; ++kt) {
        bf16x4 pb, db;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pb[r] = (bf16_t)0.0f;
            db[r] = (bf16_t)0.0f;
        }
        if constexpr (NKT > 0) {
            if (kt < NKT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = s[kt < NKT ? kt : 0][r] * inv;
                    pb[r] = f2bf(p);
                    db[r] = f2bf(p * (dp[kt < NKT ? kt : 0][r] - dl));
                }
            }
        }
        *(bf16x4*)(Pm + (q0 + (lane & 15)) * PLD + kt * 16 + (lane >> 4) * 4) = pb;
        *(bf16x4*)(Dm + (q0 + (lane & 15)) * PLD + kt * 16 + (lane >> 4) * 4) = db;
    }
    if constexpr (NKT > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dQ^T[d][q] = sum_key K[key][d] dS[q][key]
#pragma unroll
        for (int ks = 0; ks < (NKT + 1) / 2; ++ks) {
            const bf16x8 df = rfrag(Dm, PLD, q0, ks * 32, lane);
#pragma unroll
            for (int d = 0; d < 4; ++d) dq[d] = mfma16s(tfrag(Ks, SLD, ks * 32, d * 16, lane, ZR), df, dq[d]);
        }
    }
}

template <int NT, bool BIAS, int K = NT, class... A>
__device__ __forceinline__ void attn_bwd_phase1_pick(int nkt, A&&... args) {
    if (nkt == K) attn_bwd_phase1<NT, K, BIAS>(args...);
    else if constexpr (K > 0) attn_bwd_phase1_pick<NT, BIAS, K - 1>(nkt, args...);
}

// NH = (sequence, head) pairs a workgroup holds at once.  With one pair per 5-wave workgroup (78.5 KB of LDS) a CU held ONE
// workgroup, not the two the occupancy query reports (SQ PMC: 4.9 resident waves per CU), so loads, phase 1 and phase 2 ran strictly
// in turn on 5 waves over 4 SIMDs (59 us = 23 us of loads + the phases).  Two pairs in one 10-wave workgroup (NT <= 5: 2 x 78.5 KB
// in one allocation) fill the SIMDs 3 | 3 | 2 | 2; the second pair takes its tiles in reverse so that the causal triangle's costs
// add up evenly per SIMD; and the walk is persistent: the rows of the NEXT two pairs are fetched into registers (2 x 16 B per matrix
// and thread) while these are computed.  9.5 resident waves per CU, 45 us.
template <int NT, int NH, bool BIAS>
__global__ __launch_bounds__(NT * NH * 64) void attention_small_bwd_kernel(AttnBwdArgs g) {
    const AttnArgs& a = g.f;
    constexpr int KS = (NT + 1) / 2, KP = KS * 32;
    constexpr int PLD = KP + 8;          // P / dS row stride (elements): +16 B spreads 16 rows of one column over the banks (conflict ratio 0.60 -> 0.39)
    constexpr int ZR = NT * 16;          // index of the all-zero row; tiles have ZR + 1 rows
    constexpr int TR_ = ZR + 1;
    constexpr int NTH = NT * 64;         // threads per pair; NT * 16 rows x 8 pieces = 2 NTH: two pieces per thread and matrix
    // per pair: Q, K, V, dO tiles [TR_][SLD] + P, dS matrices [TR_][PLD] (rows = queries)
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave id in an SGPR: scalar branches
    const int hw = NH > 1 ? wv / NT : 0, w = wv - hw * NT;                // pair slot of this wave, wave within the slot
    const int tid = threadIdx.x - hw * NTH, lane = tid & 63;
    bf16_t* Qs = smem + hw * (4 * TR_ * SLD + 2 * TR_ * PLD);
    bf16_t* Ks = Qs + TR_ * SLD;
    bf16_t* Vs = Ks + TR_ * SLD;
    bf16_t* Os = Vs + TR_ * SLD;
    bf16_t* Pm = Os + TR_ * SLD;
    bf16_t* Dm = Pm + TR_ * PLD;
    const int qt = (hw & 1) ? NT - 1 - w : w;                // wave -> tile (phase 1: queries, phase 2: keys)
    const int total = a.B * a.H;
    bf16x8 pre[4][2];
    auto fetch = [&](int bh) {
        if (bh >= total) return;
        const int fb = bh / a.H, fh = bh % a.H;
        const int fL = a.cu ? a.cu[fb + 1] - a.cu[fb] : a.Lq;
        const size_t frow0 = a.cu ? (size_t)a.cu[fb] : (size_t)fb * a.Lq;
        const bf16_t* const bases[4] = {a.q + frow0 * a.ldq + fh * SHD, a.k + frow0 * a.ldk + fh * SHD,
                                        a.v + frow0 * a.ldv + fh * SHD, g.d_o + frow0 * g.lddo + fh * SHD};
        const int lds_[4] = {a.ldq, a.ldk, a.ldv, g.lddo};
        if (fL <= 0) return;                                 // an empty packed sequence: nothing to read, the tiles are zeroed below
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int it = 0; it < 2; ++it) {                 // unconditional loads (row clamped; rows >= L are zeroed on the way to LDS)
                const int i = tid + it * NTH, r = min(i >> 3, fL - 1), ch = i & 7;
                pre[k][it] = *(const bf16x8*)(bases[k] + (size_t)r * lds_[k] + ch * 8);
            }
    };
    fetch(blockIdx.x * NH + hw);
    for (int i = tid; i < SLD; i += NTH) {               // the zero rows: written once, no later store reaches row ZR
        Qs[ZR * SLD + i] = (bf16_t)0.0f;
        Ks[ZR * SLD + i] = (bf16_t)0.0f;
        Vs[ZR * SLD + i] = (bf16_t)0.0f;
        Os[ZR * SLD + i] = (bf16_t)0.0f;
    }
    for (int i = tid; i < PLD; i += NTH) {
        Pm[ZR * PLD + i] = (bf16_t)0.0f;
        Dm[ZR * PLD + i] = (bf16_t)0.0f;
    }
  for (int bh0 = blockIdx.x * NH; bh0 < total; bh0 += gridDim.x * NH) {      // the trip count is the same for every wave (barriers)
    const int bh = bh0 + hw;
    const bool live = bh < total;                            // an odd number of pairs leaves the last second slot idle: L = 0
    const int b = live ? bh / a.H : 0, h = live ? bh % a.H : 0;
    const int L = !live ? 0 : a.cu ? a.cu[b + 1] - a.cu[b] : a.Lq;
    const size_t row0 = a.cu ? (size_t)a.cu[b] : (size_t)b * a.Lq;
    [[maybe_unused]] const float* kbias = BIAS ? a.key_bias + (size_t)b * a.Lq : nullptr;
    __syncthreads();                                         // the previous pair's phase 2 is done with the tiles
    {
        bf16_t* const dsts[4] = {Qs, Ks, Vs, Os};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int i = tid + it * NTH, r = i >> 3, ch = i & 7;
                *(bf16x8*)(dsts[k] + r * SLD + ch * 8) = r < L ? pre[k][it] : zero8s();
            }
    }
    fetch(bh + gridDim.x * NH);
    __syncthreads();

    // ---------------- phase 1: this wave's 16 queries against all visible keys
    f32x4 dq[4];                                             // stored after the barrier, through the then dead K tile
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0, 0, 0, 0};
#if SPN_ATTN_ABL & 1
    if (false)
#endif
    {
        // Key tiles that hold a visible key for these 16 queries (wave-uniform): the causal triangle and the sequence length cut
        // them; the phase is instantiated once per count (straight-line code per wave, a scalar jump picks it)
        const int q0 = qt * 16;
        const int nkt = q0 >= L ? 0 : min(a.causal ? qt + 1 : NT, (L + 15) >> 4);
        attn_bwd_phase1_pick<NT, BIAS>(nkt, Qs, Ks, Vs, Os, Pm, Dm, q0, lane, L, a.causal, a.scale, kbias, dq);
    }
    __syncthreads();

    // ---------------- phase 2: this wave's 16 keys against all queries that see them
    {
        const int k0 = qt * 16;
        // K and V tiles are dead now: rows [k0, k0 + 16) of each are this wave's staging rows (store_rows_staged)
        store_rows_staged(Ks + k0 * SLD, SLD, dq, a.scale, lane, [&](int row) -> bf16_t* {
            return k0 + row < L ? g.dq + (row0 + k0 + row) * g.lddq + h * SHD : nullptr;
        });
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            dk[d] = f32x4{0, 0, 0, 0};
            dv[d] = f32x4{0, 0, 0, 0};
        }
        const int qs0 = a.causal ? k0 >> 5 : 0;           // first 32-query step with q >= k0
#pragma unroll
        for (int qs = 0; qs < KS; ++qs) {
            if (qs >= qs0 && !(SPN_ATTN_ABL & 2)) {
                // B operands: P^T / dS^T [k = query][j = key]  (transposed reads of the [q][key] matrices)
                const bf16x8 pf = tfrag(Pm, PLD, qs * 32, k0, lane, ZR);
                const bf16x8 df = tfrag(Dm, PLD, qs * 32, k0, lane, ZR);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    dv[d] = mfma16s(tfrag(Os, SLD, qs * 32, d * 16, lane, ZR), pf, dv[d]);   // dO^T [d][q]
                    dk[d] = mfma16s(tfrag(Qs, SLD, qs * 32, d * 16, lane, ZR), df, dk[d]);   // Q^T  [d][q]
                }
            }
        }
        store_rows_staged(Ks + k0 * SLD, SLD, dk, a.scale, lane, [&](int row) -> bf16_t* {
            return k0 + row < L ? g.dk + (row0 + k0 + row) * g.lddk + h * SHD : nullptr;
        });
        store_rows_staged(Vs + k0 * SLD, SLD, dv, 1.0f, lane, [&](int row) -> bf16_t* {
            return k0 + row < L ? g.dv + (row0 + k0 + row) * g.lddv + h * SHD : nullptr;
        });
    }
  }
}

// ------------------------------------------------------------------- few queries, many keys (forward)
// Same shape class as attention_cross_bwd_kernel: one block per (sequence, head), one wave per 16 queries (Q fragments
// in registers), keys / values streamed in tiles of 64 with the next tile's loads in flight, online softmax, P.V from
// the row-major V tile through transpose reads (no transposed LDS copy).
template <int NQ, bool BIAS>
__global__ __launch_bounds__(NQ * 64) void attention_cross_fwd_kernel(AttnArgs a) {
    constexpr int KR = 64, PLD2 = KR + 8, NTH = NQ * 64;
    constexpr int ITER = (KR * 8 + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * KR * SLD + NQ * 16 * PLD2];
    bf16_t* Ks = smem;
    bf16_t* Vs = smem + KR * SLD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    bf16_t* Ps = smem + 2 * KR * SLD + w * 16 * PLD2;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int Lq = a.Lq, Lk = a.Lk;
    const bf16_t* qb = a.q + (size_t)b * Lq * a.ldq + h * SHD;
    const bf16_t* kb = a.k + (size_t)b * Lk * a.ldk + h * SHD;
    const bf16_t* vb = a.v + (size_t)b * Lk * a.ldv + h * SHD;
    [[maybe_unused]] const float* kbias = BIAS ? a.key_bias + (size_t)b * Lk : nullptr;
    bf16x8 kreg[ITER], vreg[ITER];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTH, r = i >> 3, ch = i & 7;
            // unconditional loads of a clamped row + a select: a bounds test compiles to an exec-mask branch per load
            const int rr = min(j0 + r, Lk - 1);
            u32x4 kx = *(const u32x4*)(kb + (size_t)rr * a.ldk + ch * 8);
            u32x4 vx = *(const u32x4*)(vb + (size_t)rr * a.ldv + ch * 8);
            const bool live = (i < KR * 8) & (j0 + r < Lk);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kx[e] = live ? kx[e] : 0u;
                vx[e] = live ? vx[e] : 0u;
            }
            kreg[it] = __builtin_bit_cast(bf16x8, kx);
            vreg[it] = __builtin_bit_cast(bf16x8, vx);
        }
    };
    fetch(0);
    const int q0 = blockIdx.y * (NQ * 16) + w * 16, qrow = q0 + (lane & 15);      // grid.y: blocks of NQ * 16 queries
    const bool row_ok = qrow < Lq;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = zero8s();
        if (row_ok) qf[ks] = *(const bf16x8*)(qb + (size_t)qrow * a.ldq + ks * 32 + (lane >> 4) * 8);
    }
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;
    for (int j0 = 0; j0 < Lk; j0 += KR) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTH, r = i >> 3, ch = i & 7;
            if (i < KR * 8) {
                *(bf16x8*)(Ks + r * SLD + ch * 8) = kreg[it];
                *(bf16x8*)(Vs + r * SLD + ch * 8) = vreg[it];
            }
        }
        __syncthreads();
        if (j0 + KR < Lk) fetch(j0 + KR);
        f32x4 s[4];
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[kt] = mfma16s(rfrag(Ks, SLD, kt * 16, ks * 32, lane), qf[ks], s[kt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + kt * 16 + (lane >> 4) * 4 + r;
                float v = s[kt][r] * a.scale;
                if constexpr (BIAS) v += kbias[key < Lk ? key : Lk - 1];
                if (key >= Lk) v = -INFINITY;
                s[kt][r] = v;
                mt = fmaxf(mt, v);
            }
        }
        mt = qg_max(mt);
        const float m_new = fmaxf(m_run, mt);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bf16x4 pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m_use);
                rs += p;
                pb[r] = f2bf(p);
            }
            *(bf16x4*)(Ps + (lane & 15) * PLD2 + kt * 16 + (lane >> 4) * 4) = pb;
        }
        rs = qg_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] *= alpha;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pf = rfrag(Ps, PLD2, 0, ks * 32, lane);
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = mfma16s(tfrag(Vs, SLD, ks * 32, d * 16, lane, KR), pf, o[d]);
        }
    }
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    store_rows_staged(Ps, PLD2, o, inv, lane, [&](int row) -> bf16_t* {     // this wave's P rows are spent
        return q0 + row < Lq ? a.o + ((size_t)b * Lq + q0 + row) * a.ldo + h * SHD : nullptr;
    });
    if (!row_ok) return;
    if (a.lse && (lane >> 4) == 0)
        a.lse[((size_t)b * a.H + h) * Lq + qrow] = l_run > 0.f ? m_run + __logf(l_run) : -INFINITY;
}

// Non-causal attention with MORE than 64 queries (the ViT image towers: 257 / 577 tokens): the same kernel with 64-query
// blocks in grid.y - K / V tiles prefetched into registers and read with the LDS transpose instruction, where the tiled
// kernel of attention.hip loads each tile synchronously and transposes V with 2-byte LDS stores.
template <int NQ>
static int launch_cross_fwd_blocks(const AttnArgs& a, hipStream_t st) {
    const dim3 grid(a.B * a.H, (a.Lq + NQ * 16 - 1) / (NQ * 16));
    if (a.key_bias) hipLaunchKernelGGL((attention_cross_fwd_kernel<NQ, true>), grid, dim3(NQ * 64), 0, st, a);
    else hipLaunchKernelGGL((attention_cross_fwd_kernel<NQ, false>), grid, dim3(NQ * 64), 0, st, a);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

int attention_cross_fwd_blocks(const AttnArgs& a, hipStream_t st) {
    // waves per block (16 queries each): the fewest query blocks (every block streams all of K and V), then the fewest idle
    // waves - 257 tokens: 3 blocks of 6 waves (18 slots for 17 row tiles), 577: 5 blocks of 8
    const int tiles = (a.Lq + 15) / 16;
    int best = 4, best_blocks = (tiles + 3) / 4;
    for (int nq = 5; nq <= 8; ++nq) {
        const int nb = (tiles + nq - 1) / nq;
        if (nb < best_blocks || (nb == best_blocks && nb * nq < best_blocks * best)) { best = nq; best_blocks = nb; }
    }
    switch (best) {
        case 4: return launch_cross_fwd_blocks<4>(a, st);
        case 5: return launch_cross_fwd_blocks<5>(a, st);
        case 6: return launch_cross_fwd_blocks<6>(a, st);
        case 7: return launch_cross_fwd_blocks<7>(a, st);
        default: return launch_cross_fwd_blocks<8>(a, st);
    }
}

int attention_cross_fwd(const AttnArgs& a, hipStream_t st) {
    const dim3 grid(a.B * a.H);
    switch ((a.Lq + 15) / 16) {
#define SPN_CROSS_FWD(NQ_)                                                                                      \
    do {                                                                                                        \
        if (a.key_bias) hipLaunchKernelGGL((attention_cross_fwd_kernel<NQ_, true>), grid, dim3(NQ_ * 64), 0, st, a); \
        else hipLaunchKernelGGL((attention_cross_fwd_kernel<NQ_, false>), grid, dim3(NQ_ * 64), 0, st, a);      \
    } while (0)
        case 1: SPN_CROSS_FWD(1); break;
        case 2: SPN_CROSS_FWD(2); break;
        case 3: SPN_CROSS_FWD(3); break;
        default: SPN_CROSS_FWD(4); break;
#undef SPN_CROSS_FWD
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------ few queries, many keys (backward)
// BERT cross-attention (blip4cir/med.py:161-243): Lq <= 64 text queries over Lk = 577 image keys.  One block per
// (sequence, head) keeps Q and dO in LDS, streams the keys / values in tiles of 64 (next tile's global loads in flight
// while the current one is processed) and produces dQ (accumulated over the tiles in registers), dK and dV (complete
// per tile: all queries of the head live in this block) in ONE pass over K and V - the tiled kernels in attention.hip
// read K and V twice and re-load the query tile for every key tile.  Needs delta = rowsum(dO * O) and lse.
template <int NQ, bool BIAS>
__global__ __launch_bounds__(NQ * 64) void attention_cross_bwd_kernel(AttnBwdArgs g) {
    const AttnArgs& a = g.f;
    constexpr int QR = NQ * 16, ZRQ = QR, TRQ = QR + 1;      // query rows, zero row, rows of the query-side tiles
    constexpr int KR = 64, TRK = KR + 1;
    constexpr int PLD2 = KR + 8;                             // P / dS row stride (keys of one tile)
    constexpr int QS = (QR + 31) / 32;
    constexpr int NTH = NQ * 64;
    constexpr int ITER = (KR * 8 + NTH - 1) / NTH;           // 16-byte chunks of a K (or V) tile per thread
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* Qs = smem;
    bf16_t* Os = Qs + TRQ * SLD;
    bf16_t* Ks = Os + TRQ * SLD;
    bf16_t* Vs = Ks + TRK * SLD;
    bf16_t* Pm = Vs + TRK * SLD;
    bf16_t* Dm = Pm + TRQ * PLD2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int Lq = a.Lq, Lk = a.Lk;
    const bf16_t* qb = a.q + (size_t)b * Lq * a.ldq + h * SHD;
    const bf16_t* kb = a.k + (size_t)b * Lk * a.ldk + h * SHD;
    const bf16_t* vb = a.v + (size_t)b * Lk * a.ldv + h * SHD;
    const bf16_t* dob = g.d_o + (size_t)b * Lq * g.lddo + h * SHD;
    [[maybe_unused]] const float* kbias = BIAS ? a.key_bias + (size_t)b * Lk : nullptr;
    {
        const bf16_t* const bases[2] = {qb, dob};
        const int lds_[2] = {a.ldq, g.lddo};
        bf16_t* const dsts[2] = {Qs, Os};
        load_rows_multi<TRQ, NTH, 2>(bases, lds_, Lq, tid, dsts);
    }
    for (int i = tid; i < PLD2; i += NTH) {                  // zero rows of P / dS
        Pm[ZRQ * PLD2 + i] = (bf16_t)0.0f;
        Dm[ZRQ * PLD2 + i] = (bf16_t)0.0f;
    }
    for (int i = tid; i < SLD; i += NTH) {                   // zero rows of the key-side tiles (never addressed by tfrag
        Ks[KR * SLD + i] = (bf16_t)0.0f;                     // with zrow = KR, kept for symmetry)
        Vs[KR * SLD + i] = (bf16_t)0.0f;
    }
    // first key tile into registers
    bf16x8 kreg[ITER], vreg[ITER];
    auto fetch = [&](int j0) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTH, r = i >> 3, ch = i & 7;
            // unconditional loads of a clamped row + a select: a bounds test compiles to an exec-mask branch per load
            const int rr = min(j0 + r, Lk - 1);
            u32x4 kx = *(const u32x4*)(kb + (size_t)rr * a.ldk + ch * 8);
            u32x4 vx = *(const u32x4*)(vb + (size_t)rr * a.ldv + ch * 8);
            const bool live = (i < KR * 8) & (j0 + r < Lk);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kx[e] = live ? kx[e] : 0u;
                vx[e] = live ? vx[e] : 0u;
            }
            kreg[it] = __builtin_bit_cast(bf16x8, kx);
            vreg[it] = __builtin_bit_cast(bf16x8, vx);
        }
    };
    fetch(0);
    __syncthreads();
    const int q0 = w * 16, qrow = q0 + (lane & 15);
    const bool row_ok = qrow < Lq;
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = rfrag(Qs, SLD, q0, ks * 32, lane);
        dof[ks] = rfrag(Os, SLD, q0, ks * 32, lane);
    }
    const size_t srow = ((size_t)b * a.H + h) * Lq + (row_ok ? qrow : 0);
    const float lse = row_ok ? a.lse[srow] : INFINITY, dlt = row_ok ? g.delta[srow] : 0.f;
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0, 0, 0, 0};

    for (int j0 = 0; j0 < Lk; j0 += KR) {
        __syncthreads();                                     // the previous tile's phase B is done with Ks / Vs / Pm / Dm
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = tid + it * NTH, r = i >> 3, ch = i & 7;
            if (i < KR * 8) {
                *(bf16x8*)(Ks + r * SLD + ch * 8) = kreg[it];
                *(bf16x8*)(Vs + r * SLD + ch * 8) = vreg[it];
            }
        }
        __syncthreads();
        if (j0 + KR < Lk) fetch(j0 + KR);                    // in flight during both phases below
        // ---- phase A: this wave's 16 queries against the tile's 64 keys
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16s(rfrag(Ks, SLD, kt * 16, ks * 32, lane), qf[ks], s);
                dp = mfma16s(rfrag(Vs, SLD, kt * 16, ks * 32, lane), dof[ks], dp);
            }
            bf16x4 pb, db;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + kt * 16 + (lane >> 4) * 4 + r;
                float kb = 0.f;
                if constexpr (BIAS) kb = kbias[key < Lk ? key : Lk - 1];
                float p = __expf(s[r] * a.scale + kb - lse);             // rows past Lq: lse = +inf, p = 0 without a branch
                p = key < Lk ? p : 0.f;
                pb[r] = f2bf(p);
                db[r] = f2bf(p * (dp[r] - dlt));
            }
            *(bf16x4*)(Pm + (q0 + (lane & 15)) * PLD2 + kt * 16 + (lane >> 4) * 4) = pb;
            *(bf16x4*)(Dm + (q0 + (lane & 15)) * PLD2 + kt * 16 + (lane >> 4) * 4) = db;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dQ^T[d][q] += sum_key K[key][d] dS[q][key]   (this wave's own dS rows)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 df = rfrag(Dm, PLD2, q0, ks * 32, lane);
#pragma unroll
            for (int d = 0; d < 4; ++d) dq[d] = mfma16s(tfrag(Ks, SLD, ks * 32, d * 16, lane, KR), df, dq[d]);
        }
        __syncthreads();                                     // every wave's P / dS rows are in LDS
        // ---- phase B: 16-key subtiles of the tile, dealt round-robin to the waves
        for (int kt = w; kt < 4; kt += NQ) {
            const int k0 = kt * 16;
            f32x4 dk[4], dv[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                dk[d] = f32x4{0, 0, 0, 0};
                dv[d] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                const bf16x8 pf = tfrag(Pm, PLD2, qs * 32, k0, lane, ZRQ);
                const bf16x8 df = tfrag(Dm, PLD2, qs * 32, k0, lane, ZRQ);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    dv[d] = mfma16s(tfrag(Os, SLD, qs * 32, d * 16, lane, ZRQ), pf, dv[d]);
                    dk[d] = mfma16s(tfrag(Qs, SLD, qs * 32, d * 16, lane, ZRQ), df, dk[d]);
                }
            }
            // the K / V tiles are dead after the barrier above: rows [16 w, 16 w + 16) of each stage this wave's 16 keys
            store_rows_staged(Ks + (w * 16) * SLD, SLD, dk, a.scale, lane, [&](int row) -> bf16_t* {
                return j0 + k0 + row < Lk ? g.dk + ((size_t)b * Lk + j0 + k0 + row) * g.lddk + h * SHD : nullptr;
            });
            store_rows_staged(Vs + (w * 16) * SLD, SLD, dv, 1.0f, lane, [&](int row) -> bf16_t* {
                return j0 + k0 + row < Lk ? g.dv + ((size_t)b * Lk + j0 + k0 + row) * g.lddv + h * SHD : nullptr;
            });
        }
    }
    if (row_ok) {
        bf16_t* drow = g.dq + ((size_t)b * Lq + qrow) * g.lddq + h * SHD;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            bf16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = f2bf(dq[d][r] * a.scale);
            *(bf16x4*)(drow + d * 16 + (lane >> 4) * 4) = ob;
        }
    }
}

template <int NQ>
static int launch_cross_bwd(const AttnBwdArgs& g, hipStream_t st) {
    constexpr int TRQ = NQ * 16 + 1, TRK = 65, PLD2 = 72;
    constexpr int LDS = (2 * TRQ * SLD + 2 * TRK * SLD + 2 * TRQ * PLD2) * 2;
    if (g.f.key_bias) hipLaunchKernelGGL((attention_cross_bwd_kernel<NQ, true>), dim3(g.f.B * g.f.H), dim3(NQ * 64), LDS, st, g);
    else hipLaunchKernelGGL((attention_cross_bwd_kernel<NQ, false>), dim3(g.f.B * g.f.H), dim3(NQ * 64), LDS, st, g);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

bool attention_cross_ok(const AttnArgs& a) { return !a.causal && !a.cu && a.Lq <= 64 && a.Lk > a.Lq; }

int attention_cross_bwd(const AttnBwdArgs& g, hipStream_t st) {
    switch ((g.f.Lq + 15) / 16) {
        case 1: return launch_cross_bwd<1>(g, st);
        case 2: return launch_cross_bwd<2>(g, st);
        case 3: return launch_cross_bwd<3>(g, st);
        default: return launch_cross_bwd<4>(g, st);
    }
}

bool attention_small_ok(const AttnArgs& a) { return a.Lq == a.Lk && a.Lq <= 128; }

template <int NT>
static int launch_small_fwd(const AttnArgs& a, hipStream_t st) {
    if (a.key_bias) hipLaunchKernelGGL((attention_small_fwd_kernel<NT, true>), dim3(a.B * a.H), dim3(NT * 64), 0, st, a);
    else hipLaunchKernelGGL((attention_small_fwd_kernel<NT, false>), dim3(a.B * a.H), dim3(NT * 64), 0, st, a);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

template <int NT>
static int launch_small_bwd(const AttnBwdArgs& g, hipStream_t st) {
    constexpr int KS = (NT + 1) / 2, KP = KS * 32, PLD = KP + 8, TR_ = NT * 16 + 1;
    constexpr int NH = NT <= 5 ? 2 : 1;                      // two (sequence, head) pairs per workgroup while their tiles fit the LDS
    constexpr int LDS = NH * (4 * TR_ * SLD + 2 * TR_ * PLD) * 2;
    static_assert(LDS <= 160 * 1024, "whole-head backward tiles exceed the LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attention_small_bwd_kernel<NT, NH, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)attention_small_bwd_kernel<NT, NH, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // persistent grid: as many workgroups as are resident at once (LDS-bound), each walking its share of the pairs
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }();
    constexpr int per_cu = (160 * 1024) / LDS > 0 ? (160 * 1024) / LDS : 1;
    const int groups = (g.f.B * g.f.H + NH - 1) / NH;
    const int grid = groups < cus * per_cu ? groups : cus * per_cu;
    if (g.f.key_bias) hipLaunchKernelGGL((attention_small_bwd_kernel<NT, NH, true>), dim3(grid), dim3(NT * NH * 64), LDS, st, g);
    else hipLaunchKernelGGL((attention_small_bwd_kernel<NT, NH, false>), dim3(grid), dim3(NT * NH * 64), LDS, st, g);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

#define SPN_SMALL_DISPATCH(FN, ARG)                 \
    switch ((L + 15) / 16) {                        \
        case 1: return FN<1>(ARG, st);              \
        case 2: return FN<2>(ARG, st);              \
        case 3: return FN<3>(ARG, st);              \
        case 4: return FN<4>(ARG, st);              \
        case 5: return FN<5>(ARG, st);              \
        case 6: return FN<6>(ARG, st);              \
        case 7: return FN<7>(ARG, st);              \
        default: return FN<8>(ARG, st);             \
    }

int attention_small_fwd(const AttnArgs& a, hipStream_t st) {
    const int L = a.Lq;
    SPN_SMALL_DISPATCH(launch_small_fwd, a)
}

int attention_small_bwd(const AttnBwdArgs& g, hipStream_t st) {
    const int L = g.f.Lq;
    SPN_SMALL_DISPATCH(launch_small_bwd, g)
}

}  // namespace spn
