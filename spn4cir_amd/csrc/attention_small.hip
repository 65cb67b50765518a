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

// NT = number of 16-row tiles (= waves); keys are padded to KP = 32*ceil(NT/2) rows of zeros
template <int NT>
__global__ __launch_bounds__(NT * 64) void attention_small_fwd_kernel(AttnArgs a) {
    constexpr int KS = (NT + 1) / 2, KP = KS * 32;     // 32-deep key steps, padded key rows
    constexpr int PLD = KP + 8;                         // P row stride (elements)
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * KP * SLD + NT * 16 * PLD];
    bf16_t* Ks = smem;
    bf16_t* Vs = smem + KP * SLD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    bf16_t* Ps = smem + 2 * KP * SLD + w * 16 * PLD;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int L = a.cu ? a.cu[b + 1] - a.cu[b] : a.Lq;                 // this sequence's length
    const size_t row0 = a.cu ? (size_t)a.cu[b] : (size_t)b * a.Lq;     // its first row
    const bf16_t* qb = a.q + row0 * a.ldq + h * SHD;
    const bf16_t* kb = a.k + row0 * a.ldk + h * SHD;
    const bf16_t* vb = a.v + row0 * a.ldv + h * SHD;
    const float* kbias = a.key_bias ? a.key_bias + (size_t)b * a.Lq : nullptr;
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
    if (q0 >= L) return;
    const int nkt = a.causal ? w + 1 : NT;              // key tiles this query tile can see
    f32x4 s[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        s[kt] = f32x4{0, 0, 0, 0};
        if (kt < nkt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[kt] = mfma16s(rfrag(Ks, SLD, kt * 16, ks * 32, lane), qf[ks], s[kt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + (lane >> 4) * 4 + r;
            float v = s[kt][r] * a.scale;
            if (kbias && key < L) v += kbias[key];
            if (kt >= nkt || key >= L || (a.causal && key > qrow)) v = -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = qg_max(mx);
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
    if (qrow >= L) return;
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    bf16_t* orow = a.o + (row0 + qrow) * a.ldo + h * SHD;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        bf16x4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = f2bf(o[d][r] * inv);
        *(bf16x4*)(orow + d * 16 + (lane >> 4) * 4) = ob;
    }
    if (a.lse && (lane >> 4) == 0)
        a.lse[((size_t)b * a.H + h) * a.Lq + qrow] = sum > 0.f ? mx + __logf(sum) : -INFINITY;
}

template <int NT>
__global__ __launch_bounds__(NT * 64) void attention_small_bwd_kernel(AttnBwdArgs g) {
    const AttnArgs& a = g.f;
    constexpr int KS = (NT + 1) / 2, KP = KS * 32;
    constexpr int PLD = KP;              // P / dS row stride: key columns 0..KP-1
    constexpr int ZR = NT * 16;          // index of the all-zero row; tiles have ZR + 1 rows
    constexpr int TR_ = ZR + 1;
    // Q, K, V, dO tiles [TR_][SLD] + P, dS matrices [TR_][PLD] (rows = queries)
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* Qs = smem;
    bf16_t* Ks = Qs + TR_ * SLD;
    bf16_t* Vs = Ks + TR_ * SLD;
    bf16_t* Os = Vs + TR_ * SLD;
    bf16_t* Pm = Os + TR_ * SLD;
    bf16_t* Dm = Pm + TR_ * PLD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int L = a.cu ? a.cu[b + 1] - a.cu[b] : a.Lq;
    const size_t row0 = a.cu ? (size_t)a.cu[b] : (size_t)b * a.Lq;
    const bf16_t* qb = a.q + row0 * a.ldq + h * SHD;
    const bf16_t* kb = a.k + row0 * a.ldk + h * SHD;
    const bf16_t* vb = a.v + row0 * a.ldv + h * SHD;
    const bf16_t* dob = g.d_o + row0 * g.lddo + h * SHD;
    const float* kbias = a.key_bias ? a.key_bias + (size_t)b * a.Lq : nullptr;
    {
        const bf16_t* const bases[4] = {qb, kb, vb, dob};
        const int lds_[4] = {a.ldq, a.ldk, a.ldv, g.lddo};
        bf16_t* const dsts[4] = {Qs, Ks, Vs, Os};
        load_rows_multi<TR_, NT * 64, 4>(bases, lds_, L, tid, dsts);
    }
    for (int i = tid; i < PLD; i += NT * 64) {           // the zero row of P / dS
        Pm[ZR * PLD + i] = (bf16_t)0.0f;
        Dm[ZR * PLD + i] = (bf16_t)0.0f;
    }
    __syncthreads();

    // ---------------- phase 1: this wave's 16 queries against all visible keys
    {
        const int q0 = w * 16, qrow = q0 + (lane & 15);
        const bool row_ok = qrow < L;
        bf16x8 qf[2], dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = rfrag(Qs, SLD, q0, ks * 32, lane);
            dof[ks] = rfrag(Os, SLD, q0, ks * 32, lane);
        }
        const int nkt = a.causal ? w + 1 : NT;
        f32x4 s[NT], dp[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            s[kt] = f32x4{0, 0, 0, 0};
            dp[kt] = f32x4{0, 0, 0, 0};
            if (kt < nkt) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s[kt] = mfma16s(rfrag(Ks, SLD, kt * 16, ks * 32, lane), qf[ks], s[kt]);
                    dp[kt] = mfma16s(rfrag(Vs, SLD, kt * 16, ks * 32, lane), dof[ks], dp[kt]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + (lane >> 4) * 4 + r;
                float v = s[kt][r] * a.scale;
                if (kbias && key < L) v += kbias[key];
                if (kt >= nkt || key >= L || (a.causal && key > qrow) || !row_ok) v = -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = qg_max(mx);
        const float m_use = mx == -INFINITY ? 0.f : mx;
        float sum = 0.f, dl = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m_use);
                s[kt][r] = p;
                sum += p;
                dl += p * dp[kt][r];
            }
        sum = qg_sum(sum);
        dl = qg_sum(dl);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        dl *= inv;                                       // delta_q = sum_k P dP
#pragma unroll
        for (int kt = 0; kt < 2 * KS; ++kt) {
            bf16x4 pb, db;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = 0.f, d = 0.f;
                if (kt < NT) {
                    p = s[kt < NT ? kt : 0][r] * inv;
                    d = p * (dp[kt < NT ? kt : 0][r] - dl);
                }
                pb[r] = f2bf(p);
                db[r] = f2bf(d);
            }
            *(bf16x4*)(Pm + (q0 + (lane & 15)) * PLD + kt * 16 + (lane >> 4) * 4) = pb;
            *(bf16x4*)(Dm + (q0 + (lane & 15)) * PLD + kt * 16 + (lane >> 4) * 4) = db;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dQ^T[d][q] = sum_key K[key][d] dS[q][key]
        f32x4 dq[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d] = f32x4{0, 0, 0, 0};
        const int nks = a.causal ? (w + 2) / 2 : KS;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks < nks) {
                const bf16x8 df = rfrag(Dm, PLD, q0, ks * 32, lane);
#pragma unroll
                for (int d = 0; d < 4; ++d) dq[d] = mfma16s(tfrag(Ks, SLD, ks * 32, d * 16, lane, ZR), df, dq[d]);
            }
        }
        if (row_ok) {
            bf16_t* drow = g.dq + (row0 + qrow) * g.lddq + h * SHD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 ob;
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[r] = f2bf(dq[d][r] * a.scale);
                *(bf16x4*)(drow + d * 16 + (lane >> 4) * 4) = ob;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 2: this wave's 16 keys against all queries that see them
    {
        const int k0 = w * 16, key = k0 + (lane & 15);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            dk[d] = f32x4{0, 0, 0, 0};
            dv[d] = f32x4{0, 0, 0, 0};
        }
        const int qs0 = a.causal ? w / 2 : 0;            // first 32-query step with q >= k0
#pragma unroll
        for (int qs = 0; qs < KS; ++qs) {
            if (qs >= qs0) {
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
        if (key < L) {
            bf16_t* dkrow = g.dk + (row0 + key) * g.lddk + h * SHD;
            bf16_t* dvrow = g.dv + (row0 + key) * g.lddv + h * SHD;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 ok, ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ok[r] = f2bf(dk[d][r] * a.scale);
                    ov[r] = f2bf(dv[d][r]);
                }
                *(bf16x4*)(dkrow + d * 16 + (lane >> 4) * 4) = ok;
                *(bf16x4*)(dvrow + d * 16 + (lane >> 4) * 4) = ov;
            }
        }
    }
}

bool attention_small_ok(const AttnArgs& a) { return a.Lq == a.Lk && a.Lq <= 128; }

template <int NT>
static int launch_small_fwd(const AttnArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(attention_small_fwd_kernel<NT>, dim3(a.B * a.H), dim3(NT * 64), 0, st, a);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

template <int NT>
static int launch_small_bwd(const AttnBwdArgs& g, hipStream_t st) {
    constexpr int KS = (NT + 1) / 2, KP = KS * 32, PLD = KP, TR_ = NT * 16 + 1;
    constexpr int LDS = (4 * TR_ * SLD + 2 * TR_ * PLD) * 2;
    auto kern = attention_small_bwd_kernel<NT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.f.B * g.f.H), dim3(NT * 64), LDS, st, g);
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
