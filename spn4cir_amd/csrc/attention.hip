// Multi-head attention forward / backward, head_dim 64, bf16 MFMA with fp32 softmax.
//
// Serves every attention on the path: CLIP text (causal, L=77, clip/model.py:186-187,330-336),
// CLIP / BLIP vision (no mask, L=50..577), BLIP BERT self-attention (key padding bias,
// blip4cir/med.py:161-243) and BLIP cross-attention (Lq<=~32 text queries over 577 image keys).
//
// Flash-style: a block owns 64 query rows (4 waves x 16) and walks key tiles of 64 with an
// online softmax, so nothing of size Lq x Lk is ever written.  MFMA operands are arranged
// "swapped" (S^T = K Q^T, O^T = V^T P^T) so that each lane owns ONE query row: row statistics
// need 2 cross-lane steps, P goes back to LDS as 8-byte rows, and O is stored 8 bytes per lane.
#include "common.h"
#include "kernels.h"
#include "prof.h"

namespace spn {

static constexpr int HD = 64;        // head dim
static constexpr int KT = 64;        // keys (or queries) per LDS tile
static constexpr int LDT = 72;       // LDS row stride in elements (144 B, 16-B aligned, odd multiple of 16 B)
static constexpr int TILE_ELEMS = 64 * LDT;

__device__ __forceinline__ f32x4 mfma16a(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.0f;
    return z;
}

// Load rows [row0, row0+64) x 64 columns of a [nrows, ld] matrix (base already offset to the
// batch and head) into LDS as sN[r][c] and/or transposed sT[c][r]; rows >= nrows read as zero.
__device__ __forceinline__ void load_tile(const bf16_t* __restrict__ base, int ld, int row0, int nrows, int tid,
                                          bf16_t* sN, bf16_t* sT) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 3) + 32 * i, ch = tid & 7;
        bf16x8 v = zero8();
        if (row0 + r < nrows) v = *(const bf16x8*)(base + (size_t)(row0 + r) * ld + ch * 8);
        if (sN) *(bf16x8*)(sN + r * LDT + ch * 8) = v;
        if (sT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sT[(ch * 8 + e) * LDT + r] = v[e];
        }
    }
}

// fragment whose "row" index (lane&15) walks LDS rows and whose 8 k-values are contiguous
__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* s, int row_base, int ks, int lane) {
    return *(const bf16x8*)(s + (row_base + (lane & 15)) * LDT + ks * 32 + (lane >> 4) * 8);
}

// same, straight from global memory (rows >= nrows read as zero)
__device__ __forceinline__ bf16x8 glb_frag(const bf16_t* __restrict__ base, int ld, int row_base, int nrows, int ks,
                                           int lane) {
    const int r = row_base + (lane & 15);
    if (r >= nrows) return zero8();
    return *(const bf16x8*)(base + (size_t)r * ld + ks * 32 + (lane >> 4) * 8);
}

__device__ __forceinline__ float quad_group_max(float v) {   // across lane>>4 (lanes l, l^16, l^32, l^48)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void attention_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * TILE_ELEMS + 4 * 16 * LDT];
    bf16_t* Ks = smem;
    bf16_t* Vt = smem + TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    bf16_t* Ps = smem + 2 * TILE_ELEMS + wid * 16 * LDT;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int qblk = blockIdx.x * 64, q0 = qblk + wid * 16;
    const bf16_t* qb = a.q + (size_t)b * a.Lq * a.ldq + h * HD;
    const bf16_t* kb = a.k + (size_t)b * a.Lk * a.ldk + h * HD;
    const bf16_t* vb = a.v + (size_t)b * a.Lk * a.ldv + h * HD;
    const float* kbias = a.key_bias ? a.key_bias + (size_t)b * a.Lk : nullptr;
    const bool wave_active = q0 < a.Lq;
    const int qrow = q0 + (lane & 15);

    bf16x8 qf[2];
    qf[0] = glb_frag(qb, a.ldq, q0, a.Lq, 0, lane);
    qf[1] = glb_frag(qb, a.ldq, q0, a.Lq, 1, lane);

    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0, 0, 0, 0};
    float m_run = -INFINITY, l_run = 0.f;

    const int kend = a.causal ? min(a.Lk, qblk + 64) : a.Lk;
    for (int j0 = 0; j0 < kend; j0 += KT) {
        __syncthreads();
        load_tile(kb, a.ldk, j0, a.Lk, tid, Ks, nullptr);
        load_tile(vb, a.ldv, j0, a.Lk, tid, nullptr, Vt);
        __syncthreads();
        if (!wave_active || (a.causal && j0 > q0 + 15)) continue;

        f32x4 s[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            s[nt] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[nt] = mfma16a(lds_frag(Ks, nt * 16, ks, lane), qf[ks], s[nt]);
        }
        // s[nt][r] = S[qrow][key = j0 + nt*16 + (lane>>4)*4 + r]
        float mt = -INFINITY;
        // the key bias is fetched under ONE wave-uniform test per tile (clamped index) and the mask is a predicate: as
        // per-logit `if`s both compiled to exec-mask branches with their own waits (see attention_small.hip)
        f32x4 kb4[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) kb4[nt] = f32x4{0, 0, 0, 0};
        if (kbias) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) kb4[nt][r] = kbias[min(j0 + nt * 16 + (lane >> 4) * 4 + r, a.Lk - 1)];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + nt * 16 + (lane >> 4) * 4 + r;
                float v = s[nt][r] * a.scale + kb4[nt][r];
                const bool dead = (key >= a.Lk) | ((a.causal != 0) & (key > qrow));
                v = dead ? -INFINITY : v;
                s[nt][r] = v;
                mt = fmaxf(mt, v);
            }
        mt = quad_group_max(mt);
        const float m_new = fmaxf(m_run, mt);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float rs = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[nt][r] - m_use);
                rs += p;
                pb[r] = f2bf(p);
            }
            *(bf16x4*)(Ps + (lane & 15) * LDT + nt * 16 + (lane >> 4) * 4) = pb;
        }
        rs = quad_group_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] *= alpha;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // P rows written by this wave are visible to it
        bf16x8 pf[2];
        pf[0] = lds_frag(Ps, 0, 0, lane);
        pf[1] = lds_frag(Ps, 0, 1, lane);
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) o[d] = mfma16a(lds_frag(Vt, d * 16, ks, lane), pf[ks], o[d]);
    }
    if (!wave_active || qrow >= a.Lq) return;
    const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
    bf16_t* orow = a.o + (size_t)(b * a.Lq + qrow) * a.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        bf16x4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = f2bf(o[d][r] * inv_l);
        *(bf16x4*)(orow + d * 16 + (lane >> 4) * 4) = ob;
    }
    if (a.lse && (lane >> 4) == 0)
        a.lse[((size_t)b * a.H + h) * a.Lq + qrow] = l_run > 0.f ? m_run + __logf(l_run) : -INFINITY;
}

// SPN_ATTN_FLASH=1 forces the tiled (flash-style) kernels even where the whole-head ones apply
static bool attn_force_flash() {
    static const bool f = [] {
        const char* e = spn_env("SPN_ATTN_FLASH");
        return e && e[0] == '1';
    }();
    return f;
}

int attention_fwd(const AttnArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return SPN_ERR_ARG;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 4) return SPN_ERR_SHAPE;
    if (a.cu && !attention_small_ok(a)) return SPN_ERR_SHAPE;   // packed rows: whole-head kernels only
    if (attention_small_ok(a) && (a.cu || !attn_force_flash())) {
        ProfScope prof(PK_ATTN_FWD, 4.0 * a.B * a.H * a.Lq * a.Lk * HD, st);
        return attention_small_fwd(a, st);
    }
    if (attention_cross_ok(a) && !attn_force_flash()) {
        ProfScope prof(PK_ATTN_FWD, 4.0 * a.B * a.H * a.Lq * a.Lk * HD, st);
        return attention_cross_fwd(a, st);
    }
    if (!a.causal && !a.cu && !attn_force_flash()) {       // the image towers (257 / 577 tokens)
        ProfScope prof(PK_ATTN_FWD, 4.0 * a.B * a.H * a.Lq * a.Lk * HD, st);
        return attention_cross_fwd_blocks(a, st);
    }
    {
        ProfScope prof(PK_ATTN_FWD, 4.0 * a.B * a.H * a.Lq * a.Lk * HD, st);
        hipLaunchKernelGGL(attention_fwd_kernel, dim3((a.Lq + 63) / 64, a.B * a.H), dim3(256), 0, st, a);
    }
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

// ------------------------------------------------------------------------------ backward
// delta[b,h,q] = sum_d dO[q,d] * O[q,d]
__global__ void attention_delta_kernel(const bf16_t* __restrict__ o, int ldo, const bf16_t* __restrict__ d_o, int lddo,
                                       float* __restrict__ delta, int B, int H, int Lq) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over B*Lq*H, h fastest (coalesced rows)
    if (idx >= B * Lq * H) return;
    const int h = idx % H, row = idx / H;                    // row = b*Lq + q
    const bf16_t* po = o + (size_t)row * ldo + h * HD;
    const bf16_t* pd = d_o + (size_t)row * lddo + h * HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bf16x8 x = *(const bf16x8*)(po + c * 8), y = *(const bf16x8*)(pd + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
    }
    const int b = row / Lq, q = row % Lq;
    delta[((size_t)b * H + h) * Lq + q] = s;
}

// dQ: block = 64 query rows, walks key tiles.  Per lane one query row (as in the forward).
__global__ __launch_bounds__(256) void attention_bwd_dq_kernel(AttnBwdArgs g) {
    const AttnArgs& a = g.f;
    __shared__ __attribute__((aligned(16))) bf16_t smem[3 * TILE_ELEMS + 4 * 16 * LDT];
    bf16_t* Ks = smem;
    bf16_t* Kt = smem + TILE_ELEMS;
    bf16_t* Vs = smem + 2 * TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    bf16_t* Ds = smem + 3 * TILE_ELEMS + wid * 16 * LDT;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int qblk = blockIdx.x * 64, q0 = qblk + wid * 16;
    const bf16_t* qb = a.q + (size_t)b * a.Lq * a.ldq + h * HD;
    const bf16_t* kb = a.k + (size_t)b * a.Lk * a.ldk + h * HD;
    const bf16_t* vb = a.v + (size_t)b * a.Lk * a.ldv + h * HD;
    const bf16_t* dob = g.d_o + (size_t)b * a.Lq * g.lddo + h * HD;
    const float* kbias = a.key_bias ? a.key_bias + (size_t)b * a.Lk : nullptr;
    const bool wave_active = q0 < a.Lq;
    const int qrow = q0 + (lane & 15);
    const bool row_ok = qrow < a.Lq;
    const size_t srow = ((size_t)b * a.H + h) * a.Lq + (row_ok ? qrow : 0);
    const float lse = row_ok ? a.lse[srow] : 0.f;
    const float dlt = row_ok ? g.delta[srow] : 0.f;

    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = glb_frag(qb, a.ldq, q0, a.Lq, ks, lane);
        dof[ks] = glb_frag(dob, g.lddo, q0, a.Lq, ks, lane);
    }
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0, 0, 0, 0};

    const int kend = a.causal ? min(a.Lk, qblk + 64) : a.Lk;
    for (int j0 = 0; j0 < kend; j0 += KT) {
        __syncthreads();
        load_tile(kb, a.ldk, j0, a.Lk, tid, Ks, Kt);
        load_tile(vb, a.ldv, j0, a.Lk, tid, Vs, nullptr);
        __syncthreads();
        if (!wave_active || (a.causal && j0 > q0 + 15)) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16a(lds_frag(Ks, nt * 16, ks, lane), qf[ks], s);
                dp = mfma16a(lds_frag(Vs, nt * 16, ks, lane), dof[ks], dp);
            }
            bf16x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + nt * 16 + (lane >> 4) * 4 + r;
                float v = s[r] * a.scale;
                if (kbias) v += kbias[min(key, a.Lk - 1)];                       // wave-uniform test, clamped index
                const bool masked = (key >= a.Lk) | ((a.causal != 0) & (key > qrow)) | !row_ok;
                float p = __expf(v - lse);
                p = masked ? 0.f : p;
                ds[r] = f2bf(p * (dp[r] - dlt));
            }
            *(bf16x4*)(Ds + (lane & 15) * LDT + nt * 16 + (lane >> 4) * 4) = ds;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8 dsf[2];
        dsf[0] = lds_frag(Ds, 0, 0, lane);
        dsf[1] = lds_frag(Ds, 0, 1, lane);
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dq[d] = mfma16a(lds_frag(Kt, d * 16, ks, lane), dsf[ks], dq[d]);
    }
    if (!wave_active || !row_ok) return;
    bf16_t* drow = g.dq + (size_t)(b * a.Lq + qrow) * g.lddq + h * HD;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        bf16x4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[r] = f2bf(dq[d][r] * a.scale);
        *(bf16x4*)(drow + d * 16 + (lane >> 4) * 4) = ob;
    }
}

// dK, dV: block = 64 keys (4 waves x 16), walks query tiles.  Per lane one key.
__global__ __launch_bounds__(256) void attention_bwd_dkv_kernel(AttnBwdArgs g) {
    const AttnArgs& a = g.f;
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE_ELEMS + 2 * 4 * 16 * LDT];
    __shared__ float s_lse[64], s_dlt[64];
    bf16_t* Qs = smem;
    bf16_t* Qt = smem + TILE_ELEMS;
    bf16_t* dOs = smem + 2 * TILE_ELEMS;
    bf16_t* dOt = smem + 3 * TILE_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    bf16_t* Pw = smem + 4 * TILE_ELEMS + wid * 16 * LDT;           // [key][q] probabilities
    bf16_t* Dw = smem + 4 * TILE_ELEMS + (4 + wid) * 16 * LDT;     // [key][q] dS
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int kblk = blockIdx.x * 64, k0 = kblk + wid * 16;
    const bf16_t* qb = a.q + (size_t)b * a.Lq * a.ldq + h * HD;
    const bf16_t* kb = a.k + (size_t)b * a.Lk * a.ldk + h * HD;
    const bf16_t* vb = a.v + (size_t)b * a.Lk * a.ldv + h * HD;
    const bf16_t* dob = g.d_o + (size_t)b * a.Lq * g.lddo + h * HD;
    const bool wave_active = k0 < a.Lk;
    const int key = k0 + (lane & 15);
    const bool key_ok = key < a.Lk;
    const float kbias = (a.key_bias && key_ok) ? a.key_bias[(size_t)b * a.Lk + key] : 0.f;

    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = glb_frag(kb, a.ldk, k0, a.Lk, ks, lane);
        vf[ks] = glb_frag(vb, a.ldv, k0, a.Lk, ks, lane);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        dk[d] = f32x4{0, 0, 0, 0};
        dv[d] = f32x4{0, 0, 0, 0};
    }
    const int qstart = a.causal ? (kblk / KT) * KT : 0;   // queries before the first key of the block see none of it
    for (int i0 = qstart; i0 < a.Lq; i0 += KT) {
        __syncthreads();
        load_tile(qb, a.ldq, i0, a.Lq, tid, Qs, Qt);
        load_tile(dob, g.lddo, i0, a.Lq, tid, dOs, dOt);
        if (tid < 64) {
            const int qi = i0 + tid;
            const size_t sr = ((size_t)b * a.H + h) * a.Lq + (qi < a.Lq ? qi : 0);
            s_lse[tid] = qi < a.Lq ? a.lse[sr] : 0.f;
            s_dlt[tid] = qi < a.Lq ? g.delta[sr] : 0.f;
        }
        __syncthreads();
        if (!wave_active || (a.causal && i0 + 63 < k0)) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            // s[r] = S[q = i0 + nt*16 + (lane>>4)*4 + r][key]
            f32x4 s = {0, 0, 0, 0}, dp = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16a(lds_frag(Qs, nt * 16, ks, lane), kf[ks], s);
                dp = mfma16a(lds_frag(dOs, nt * 16, ks, lane), vf[ks], dp);
            }
            bf16x4 pb, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = nt * 16 + (lane >> 4) * 4 + r, qi = i0 + ql;
                const bool masked = !key_ok || qi >= a.Lq || (a.causal && key > qi);
                const float p = masked ? 0.f : __expf(s[r] * a.scale + kbias - s_lse[ql]);
                pb[r] = f2bf(p);
                ds[r] = f2bf(p * (dp[r] - s_dlt[ql]));
            }
            *(bf16x4*)(Pw + (lane & 15) * LDT + nt * 16 + (lane >> 4) * 4) = pb;
            *(bf16x4*)(Dw + (lane & 15) * LDT + nt * 16 + (lane >> 4) * 4) = ds;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            pf[ks] = lds_frag(Pw, 0, ks, lane);
            dsf[ks] = lds_frag(Dw, 0, ks, lane);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                dv[d] = mfma16a(lds_frag(dOt, d * 16, ks, lane), pf[ks], dv[d]);
                dk[d] = mfma16a(lds_frag(Qt, d * 16, ks, lane), dsf[ks], dk[d]);
            }
    }
    if (!wave_active || !key_ok) return;
    bf16_t* dkrow = g.dk + (size_t)(b * a.Lk + key) * g.lddk + h * HD;
    bf16_t* dvrow = g.dv + (size_t)(b * a.Lk + key) * g.lddv + h * HD;
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

int attention_bwd(const AttnBwdArgs& g, hipStream_t st) {
    const AttnArgs& a = g.f;
    if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0 || !a.lse || !g.delta) return SPN_ERR_ARG;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 8 || g.lddo % 8 || g.lddq % 4 || g.lddk % 4 || g.lddv % 4)
        return SPN_ERR_SHAPE;
    const int n = a.B * a.Lq * a.H;
    ProfScope prof(PK_ATTN_BWD, 10.0 * a.B * a.H * a.Lq * a.Lk * HD, st);
    if (a.cu && !attention_small_ok(a)) return SPN_ERR_SHAPE;
    if (attention_small_ok(a) && (a.cu || !attn_force_flash())) return attention_small_bwd(g, st);
    hipLaunchKernelGGL(attention_delta_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const bf16_t*)a.o, a.ldo, g.d_o,
                       g.lddo, g.delta, a.B, a.H, a.Lq);
    SPN_CHECK_LAUNCH();
    if (attention_cross_ok(a) && !attn_force_flash()) return attention_cross_bwd(g, st);
    hipLaunchKernelGGL(attention_bwd_dq_kernel, dim3((a.Lq + 63) / 64, a.B * a.H), dim3(256), 0, st, g);
    SPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(attention_bwd_dkv_kernel, dim3((a.Lk + 63) / 64, a.B * a.H), dim3(256), 0, st, g);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
