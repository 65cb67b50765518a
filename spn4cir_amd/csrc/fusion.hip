// BLIP multimodal text encoder (blip4cir/med.py BertModel, mode='multimodal') forward + backward and the
// text_proj / normalise head of blip_cir.py:98 - the query producer of blip4cir's bank step
// (blip4cir/models.py:95-105).  Post-LN BERT layers: self-attention (padding bias), cross-attention over
// the reference image's tokens (K/V = Linear(enc_width -> W) of the image tokens, per layer), exact-GELU
// FFN.  Pure launch sequencing over the library's kernels; the q/k/v Linears of a block are packed into
// one GEMM (rows ordered query, key, value).  The image tokens are constants (frozen ViT, detached bank):
// no gradient flows into them.
#include "tower.h"

namespace spn {

#define SPN_TRYF(x)                       \
    do {                                  \
        int rc__ = (x);                   \
        if (rc__ != SPN_OK) return rc__;  \
    } while (0)

static inline size_t fa(size_t x) { return (x + 255) & ~(size_t)255; }

// per-layer parameter order (fp32 flat); LO_* index FusionLayout::layer_off
enum {
    LO_SA_WQKV = 0, LO_SA_BQKV, LO_SA_WO, LO_SA_BO, LO_SA_LNG, LO_SA_LNB,
    LO_CA_WQ, LO_CA_BQ, LO_CA_WKV, LO_CA_BKV, LO_CA_WO, LO_CA_BO, LO_CA_LNG, LO_CA_LNB,
    LO_FF_W1, LO_FF_B1, LO_FF_W2, LO_FF_B2, LO_FF_LNG, LO_FF_LNB, LO_SIZE
};
// per-layer bf16 mirror order (each weight followed by its transpose); BO_* index FusionLayout::bf16_off
enum { BO_SA_WQKV = 0, BO_SA_WQKV_T, BO_SA_WO, BO_SA_WO_T, BO_CA_WQ, BO_CA_WQ_T, BO_CA_WKV, BO_CA_WKV_T, BO_CA_WO,
       BO_CA_WO_T, BO_FF_W1, BO_FF_W1_T, BO_FF_W2, BO_FF_W2_T, BO_SIZE };

void fusion_layout(const FusionCfg& c, FusionLayout* t) {
    const int64_t W = c.W, E = c.E, I = c.I;
    int64_t o = 0;
    t->word = o; o += (int64_t)c.vocab * W;
    t->pos = o; o += (int64_t)c.max_pos * W;
    t->emb_ln_g = o; o += W;
    t->emb_ln_b = o; o += W;
    t->layers = o;
    const int64_t sz[LO_SIZE] = {3 * W * W, 3 * W, W * W, W, W, W, W * W, W, 2 * W * E, 2 * W, W * W, W, W, W,
                                 I * W, I, W * I, W, W, W};
    int64_t lo = 0;
    for (int i = 0; i < LO_SIZE; ++i) { t->layer_off[i] = lo; lo += sz[i]; }
    t->layer_off[LO_SIZE] = lo;
    t->layer_size = lo;
    o += lo * c.layers;
    t->proj_w = o; o += (int64_t)c.Dp * W;
    t->proj_b = o; o += c.Dp;
    t->n_params = o;
    const int64_t bs[BO_SIZE / 2] = {3 * W * W, W * W, W * W, 2 * W * E, W * W, I * W, W * I};
    int64_t bo = 0;
    for (int i = 0; i < BO_SIZE / 2; ++i) {
        t->bf16_off[2 * i] = bo; bo += bs[i];
        t->bf16_off[2 * i + 1] = bo; bo += bs[i];
    }
    t->bf16_off[BO_SIZE] = bo;
    t->bf16_layer_size = bo;
    t->bf16_proj = bo * c.layers;
    t->bf16_proj_t = t->bf16_proj + (int64_t)c.Dp * W;
    t->n_bf16 = t->bf16_proj_t + (int64_t)c.Dp * W;
}

// Cross-attention with the K/V projections absorbed into the query / output side (xattn.hip) whenever the shape allows it:
// the activation and workspace carve-ups below depend on it, so it is a pure function of the configuration (+ SPN_XATTN_ABSORB).
static bool fusion_absorb(const FusionCfg& c) { return xattn_absorb_ok(c.B, c.L, c.H, c.S, c.E, c.W); }
// Text rows of the batch: B*L padded positions, or - packed (c.T > 0, the caller passes the prefix sums of the caption lengths
// where the dense form takes the attention mask) - only the T unmasked ones.  A padded position influences neither the [ENC]
// feature nor any gradient (its key is masked in every self-attention, med.py:686), so dropping the rows changes no result.
static int fusion_rows(const FusionCfg& c) { return c.T > 0 ? c.T : c.B * c.L; }
static size_t fusion_the(const FusionCfg& c) { return (size_t)fusion_rows(c) * c.H * c.E; }                // elements of [T, H, E]
static size_t fusion_rsp(const FusionCfg& c) { return (size_t)fusion_rows(c) * c.H * xattn_sp(c.S); }      // elements of [T*H, SP]

// Pooled last layer (SPN_POOL_LAST=0 switches it off): blip_cir.py:98 reads ONE row of the encoder output - the [ENC] position -
// so of the LAST layer only that row is needed, and everything behind its self-attention is row-wise or attends over the image
// tokens only: the self-attention output projection, the whole cross-attention (H query rows per sample), both LayerNorms and
// the FFN run on the B [ENC] rows; the backward scatters the two gradients that re-enter the all-rows part (residual stream,
// self-attention context) and continues there.  Needs the absorbed cross-attention (its kernels take any row count per sample).
static bool fusion_pool_last(const FusionCfg& c) {
    static const bool off = [] {
        const char* e = spn_env("SPN_POOL_LAST");
        return e && e[0] == '0';
    }();
    return !off && fusion_absorb(c);
}

int fusion_packed_ok(const FusionCfg& c) { return c.L <= 128 && fusion_absorb(c) ? 1 : 0; }

static int fusion_check(const FusionCfg& c) {
    if (c.B <= 0 || c.L <= 0 || c.S <= 0 || c.layers <= 0 || c.L > c.max_pos) return SPN_ERR_ARG;
    if (c.W % 64 || c.H * 64 != c.W || c.E % 64 || c.I % 64 || c.Dp % 64) return SPN_ERR_SHAPE;
    if (c.T < 0 || (c.T > 0 && (c.T < c.B || (int64_t)c.T > (int64_t)c.B * c.L))) return SPN_ERR_ARG;
    // packed rows: whole-head self-attention kernels (L <= 128) and the absorbed cross-attention (per-sample row ranges)
    if (c.T > 0 && (c.L > 128 || !fusion_absorb(c))) return SPN_ERR_SHAPE;
    return SPN_OK;
}

int fusion_refresh_bf16(const FusionCfg& c, const float* params, bf16_t* wb, hipStream_t st) {
    SPN_TRYF(fusion_check(c));
    FusionLayout t;
    fusion_layout(c, &t);
    const int W = c.W, E = c.E, I = c.I;
    const int src[BO_SIZE / 2] = {LO_SA_WQKV, LO_SA_WO, LO_CA_WQ, LO_CA_WKV, LO_CA_WO, LO_FF_W1, LO_FF_W2};
    const int rows[BO_SIZE / 2] = {3 * W, W, W, 2 * W, W, I, W};
    const int cols[BO_SIZE / 2] = {W, W, W, E, W, W, I};
    // all layers at once, four matrices per launch (cast_transpose_multi) instead of one launch per matrix
    for (int i0 = 0; i0 < BO_SIZE / 2; i0 += 4) {
        CastTransposeSet d{};
        d.tile_start[0] = 0;
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k;
            if (i < BO_SIZE / 2) {
                d.src[k] = t.layer_off[src[i]]; d.dst[k] = t.bf16_off[2 * i]; d.dst_t[k] = t.bf16_off[2 * i + 1];
                d.rows[k] = rows[i]; d.cols[k] = cols[i];
                d.tile_start[k + 1] = d.tile_start[k] + ((rows[i] + 31) / 32) * ((cols[i] + 31) / 32);
            } else {
                d.rows[k] = d.cols[k] = 0;
                d.tile_start[k + 1] = d.tile_start[k];
            }
        }
        d.p_stride = t.layer_size; d.wb_stride = t.bf16_layer_size;
        SPN_TRYF(cast_transpose_multi(params + t.layers, wb, d, c.layers, st));
    }
    SPN_TRYF(cast_transpose_f32_bf16(params + t.proj_w, wb + t.bf16_proj, wb + t.bf16_proj_t, c.Dp, c.W, st));
    return SPN_OK;
}

// ------------------------------------------------------------------------------ activations
struct FusionLayerActs {
    float* x_in; bf16_t* xb_in;                 // layer input (previous LN output), fp32 + bf16
    bf16_t* qkv; float* lse1; bf16_t* ctx1; float* y1; float *mean1, *rstd1; float* x1; bf16_t* x1b;
    bf16_t* q2; bf16_t* kv2; float* lse2; bf16_t* ctx2; float* y2; float *mean2, *rstd2; float* x2; bf16_t* x2b;
    bf16_t *qa, *pm, *oa;                       // absorbed cross-attention: Q' [T,H,E], P [B,R,SP], O' [T,H,E] (then kv2 = lse2 = null)
    bf16_t* pre; bf16_t* u; float* y3; float *mean3, *rstd3;
};

struct FusionActs {
    float* key_bias;        // [B, L]   (1 - mask) * -10000
    int32_t* last;          // [B]      index of the last unmasked token (embedding backward)
    int32_t* zero_idx;      // [B]      zeros: row of the [ENC] token
    int32_t *cu, *row_b, *row_l;   // packed: prefix sums [B+1] (cu[b] = the [ENC] row of sample b), sample / position of each row [T]
    bf16_t* enc_b;          // [B*S, E] bf16 copy of the image tokens
    float* emb;             // [T, W]   word + position (LN input)
    float *emb_mean, *emb_rstd;
    char* layers; size_t layer_bytes;
    float* x_final; bf16_t* xb_final;           // output of the last layer
    float* h0; bf16_t* h0b;                     // [B, W] the [ENC] rows
    float* pool_x; bf16_t* pool_ctx;            // [B, W] pooled last layer: residual stream / self-attention context at the [ENC] rows
    float* proj;                                // [B, Dp] text_proj output (pre-normalise)
};

static size_t fusion_layer_act_bytes(const FusionCfg& c) {
    const size_t T = (size_t)fusion_rows(c), TS = (size_t)c.B * c.S, W = c.W, I = c.I;
    size_t b = 0;
    b += fa(T * W * 4) + fa(T * W * 2);                                        // x_in, xb_in
    b += fa(T * 3 * W * 2) + fa((size_t)c.B * c.H * c.L * 4) + fa(T * W * 2);   // qkv, lse1, ctx1
    b += fa(T * W * 4) + 2 * fa(T * 4) + fa(T * W * 4) + fa(T * W * 2);         // y1, mean1, rstd1, x1, x1b
    if (fusion_absorb(c)) b += fa(T * W * 2) + 2 * fa(fusion_the(c) * 2) + fa(fusion_rsp(c) * 2) + fa(T * W * 2);   // q2, qa, oa, pm, ctx2
    else b += fa(T * W * 2) + fa(TS * 2 * W * 2) + fa((size_t)c.B * c.H * c.L * 4) + fa(T * W * 2);   // q2, kv2, lse2, ctx2
    b += fa(T * W * 4) + 2 * fa(T * 4) + fa(T * W * 4) + fa(T * W * 2);         // y2, mean2, rstd2, x2, x2b
    b += 2 * fa(T * I * 2) + fa(T * W * 4) + 2 * fa(T * 4);                     // pre, u, y3, mean3, rstd3
    return b;
}

static FusionLayerActs fusion_layer_acts_at(char* base, const FusionCfg& c) {
    const size_t T = (size_t)fusion_rows(c), TS = (size_t)c.B * c.S, W = c.W, I = c.I;
    char* p = base;
    auto take = [&](size_t bytes) { char* r = p; p += fa(bytes); return r; };
    FusionLayerActs A;
    A.x_in = (float*)take(T * W * 4); A.xb_in = (bf16_t*)take(T * W * 2);
    A.qkv = (bf16_t*)take(T * 3 * W * 2); A.lse1 = (float*)take((size_t)c.B * c.H * c.L * 4); A.ctx1 = (bf16_t*)take(T * W * 2);
    A.y1 = (float*)take(T * W * 4); A.mean1 = (float*)take(T * 4); A.rstd1 = (float*)take(T * 4);
    A.x1 = (float*)take(T * W * 4); A.x1b = (bf16_t*)take(T * W * 2);
    A.q2 = (bf16_t*)take(T * W * 2);
    A.kv2 = nullptr; A.lse2 = nullptr; A.qa = A.pm = A.oa = nullptr;
    if (fusion_absorb(c)) {
        A.qa = (bf16_t*)take(fusion_the(c) * 2); A.oa = (bf16_t*)take(fusion_the(c) * 2); A.pm = (bf16_t*)take(fusion_rsp(c) * 2);
    } else {
        A.kv2 = (bf16_t*)take(TS * 2 * W * 2); A.lse2 = (float*)take((size_t)c.B * c.H * c.L * 4);
    }
    A.ctx2 = (bf16_t*)take(T * W * 2);
    A.y2 = (float*)take(T * W * 4); A.mean2 = (float*)take(T * 4); A.rstd2 = (float*)take(T * 4);
    A.x2 = (float*)take(T * W * 4); A.x2b = (bf16_t*)take(T * W * 2);
    A.pre = (bf16_t*)take(T * I * 2); A.u = (bf16_t*)take(T * I * 2);
    A.y3 = (float*)take(T * W * 4); A.mean3 = (float*)take(T * 4); A.rstd3 = (float*)take(T * 4);
    return A;
}

size_t fusion_act_bytes(const FusionCfg& c) {
    const size_t T = (size_t)fusion_rows(c), TS = (size_t)c.B * c.S, W = c.W;
    size_t b = fa(T * 4) + 2 * fa((size_t)c.B * 4) + fa(TS * c.E * 2) + fa(T * W * 4) + 2 * fa(T * 4);
    // cu / row_b / row_l of the packed form.  The dense size (T = 0) reserves them for the largest packed batch (T = B * L) too, so
    // that an arena sized with T = 0 really fits EVERY batch of the shape: each other term is monotone in the row count
    b += fa((size_t)(c.B + 1) * 4) + 2 * fa(T * 4);
    b += fusion_layer_act_bytes(c) * c.layers;
    b += fa(T * W * 4) + fa(T * W * 2) + fa((size_t)c.B * W * 4) + fa((size_t)c.B * W * 2) + fa((size_t)c.B * c.Dp * 4);
    b += fa((size_t)c.B * W * 4) + fa((size_t)c.B * W * 2);          // pool_x, pool_ctx
    return b;
}

static FusionActs fusion_acts_at(char* base, const FusionCfg& c) {
    const size_t T = (size_t)fusion_rows(c), TS = (size_t)c.B * c.S, W = c.W;
    char* p = base;
    auto take = [&](size_t bytes) { char* r = p; p += fa(bytes); return r; };
    FusionActs A;
    A.key_bias = (float*)take(T * 4);
    A.last = (int32_t*)take((size_t)c.B * 4);
    A.zero_idx = (int32_t*)take((size_t)c.B * 4);
    A.cu = A.row_b = A.row_l = nullptr;
    if (c.T > 0) {
        A.cu = (int32_t*)take((size_t)(c.B + 1) * 4); A.row_b = (int32_t*)take(T * 4); A.row_l = (int32_t*)take(T * 4);
    }
    A.enc_b = (bf16_t*)take(TS * c.E * 2);
    A.emb = (float*)take(T * W * 4);
    A.emb_mean = (float*)take(T * 4);
    A.emb_rstd = (float*)take(T * 4);
    A.layer_bytes = fusion_layer_act_bytes(c);
    A.layers = p; p += A.layer_bytes * c.layers;
    A.x_final = (float*)take(T * W * 4);
    A.xb_final = (bf16_t*)take(T * W * 2);
    A.h0 = (float*)take((size_t)c.B * W * 4);
    A.h0b = (bf16_t*)take((size_t)c.B * W * 2);
    A.proj = (float*)take((size_t)c.B * c.Dp * 4);
    A.pool_x = (float*)take((size_t)c.B * W * 4);
    A.pool_ctx = (bf16_t*)take((size_t)c.B * W * 2);
    return A;
}

// key_bias[b,l] = (1 - mask) * -10000 (med.py:686); last[b] = (number of unmasked tokens) - 1; zero_idx = 0
__global__ void fusion_mask_kernel(const int32_t* __restrict__ mask, float* __restrict__ key_bias,
                                   int32_t* __restrict__ last, int32_t* __restrict__ zero_idx, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int l = 0; l < L; ++l) {
        const int m = mask ? mask[(size_t)b * L + l] : 1;
        key_bias[(size_t)b * L + l] = m ? 0.f : -10000.0f;
        if (m) n = l + 1;
    }
    last[b] = n > 0 ? n - 1 : 0;
    zero_idx[b] = 0;
}

static int nt(const bf16_t* A, const bf16_t* Bw, int M, int N, int K, const float* bias, bf16_t* out_b, float* out_f,
              hipStream_t st, int act = ACT_NONE, bf16_t* pre = nullptr) {
    GemmEpilogue e;
    e.bias = bias; e.out_bf16 = out_b; e.out_f32 = out_f; e.ldc = N; e.act = act; e.aux_out = pre; e.aux_grad = 1;
    return gemm_nt(A, Bw, M, N, K, K, K, GEMM_STORE, e, st);
}

static int nt_resid(const bf16_t* A, const bf16_t* Bw, int M, int N, int K, const float* bias, const float* resid,
                    float* out_f, hipStream_t st) {
    GemmEpilogue e;
    e.bias = bias; e.resid = resid; e.ldr = N; e.out_f32 = out_f; e.ldc = N;
    return gemm_nt(A, Bw, M, N, K, K, K, GEMM_RESID, e, st);
}

static AttnArgs self_attn_args(const FusionCfg& c, const FusionLayerActs& A, const float* key_bias, const int32_t* cu) {
    AttnArgs a;
    a.q = A.qkv; a.k = A.qkv + c.W; a.v = A.qkv + 2 * c.W;
    a.ldq = a.ldk = a.ldv = 3 * c.W;
    a.o = A.ctx1; a.ldo = c.W; a.lse = A.lse1; a.key_bias = cu ? nullptr : key_bias; a.cu = cu;
    a.B = c.B; a.H = c.H; a.Lq = c.L; a.Lk = c.L; a.causal = 0; a.scale = 0.125f;
    return a;
}

static AttnArgs cross_attn_args(const FusionCfg& c, const FusionLayerActs& A) {
    AttnArgs a;
    a.q = A.q2; a.k = A.kv2; a.v = A.kv2 + c.W;
    a.ldq = c.W; a.ldk = a.ldv = 2 * c.W;
    a.o = A.ctx2; a.ldo = c.W; a.lse = A.lse2; a.key_bias = nullptr;     // image mask is all ones (blip_cir.py:85)
    a.B = c.B; a.H = c.H; a.Lq = c.L; a.Lk = c.S; a.causal = 0; a.scale = 0.125f;
    return a;
}

int fusion_fwd(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, const int32_t* mask,
               const float* enc, char* acts, float* proj_out, hipStream_t st, const bf16_t* token_bank, const int64_t* token_idx,
               int64_t bank_rows) {
    SPN_TRYF(fusion_check(c));
    FusionLayout t;
    fusion_layout(c, &t);
    FusionActs A = fusion_acts_at(acts, c);
    const int T = fusion_rows(c), TS = c.B * c.S, W = c.W, I = c.I, E = c.E;
    const bool absorb = fusion_absorb(c), pool_last = fusion_pool_last(c);
    const bool packed = c.T > 0;
    if (packed) {
        if (!mask) return SPN_ERR_ARG;                   // packed: the mask argument carries cu_seqlens int32 [B + 1]
        SPN_TRYF(build_row_map(mask, A.row_b, A.row_l, A.last, c.B, st, A.cu));
    } else {
        hipLaunchKernelGGL(fusion_mask_kernel, dim3((c.B + 63) / 64), dim3(64), 0, st, mask, A.key_bias, A.last, A.zero_idx, c.B,
                           c.L);
        SPN_CHECK_LAUNCH();
    }
    // image tokens of the batch as the bf16 A operand of the K/V projections: cast of a caller-gathered fp32 block, or gathered
    // here from a device-resident bf16 token bank (row = one image's S x E tokens; blip4cir/models.py:97-100)
    if (token_bank) SPN_TRYF(gather_bank_rows_bf16(token_bank, token_idx, bank_rows, A.enc_b, c.B, (size_t)c.S * E, st));
    else SPN_TRYF(cast_f32_bf16(enc, A.enc_b, (size_t)TS * E, st));
    if (packed) SPN_TRYF(embed_fwd_packed(ids, A.row_b, A.row_l, params + t.word, params + t.pos, A.emb, T, c.L, W, c.vocab, st));
    else SPN_TRYF(embed_fwd(ids, params + t.word, params + t.pos, A.emb, c.B, c.L, W, c.vocab, st));
    FusionLayerActs first = fusion_layer_acts_at(A.layers, c);
    SPN_TRYF(layernorm_fwd(A.emb, params + t.emb_ln_g, params + t.emb_ln_b, first.xb_in, first.x_in, A.emb_mean, A.emb_rstd, T,
                           W, 1e-12f, st));
    for (int l = 0; l < c.layers; ++l) {
        FusionLayerActs a = fusion_layer_acts_at(A.layers + A.layer_bytes * l, c);
        float* x_out = A.x_final;
        bf16_t* xb_out = A.xb_final;
        if (l + 1 < c.layers) {
            FusionLayerActs n = fusion_layer_acts_at(A.layers + A.layer_bytes * (l + 1), c);
            x_out = n.x_in; xb_out = n.xb_in;
        }
        const float* p = params + t.layers + t.layer_size * l;
        const bf16_t* b = wb + t.bf16_layer_size * l;
        auto P = [&](int i) { return p + t.layer_off[i]; };
        auto Bw = [&](int i) { return b + t.bf16_off[i]; };
        // self-attention (med.py BertSelfAttention + BertSelfOutput)
        SPN_TRYF(nt(a.xb_in, Bw(BO_SA_WQKV), T, 3 * W, W, P(LO_SA_BQKV), a.qkv, nullptr, st));
        SPN_TRYF(attention_fwd(self_attn_args(c, a, A.key_bias, A.cu), st));
        // from here on a layer is row-wise (or attends over the image tokens): the pooled last layer continues on the [ENC] rows,
        // one per sample, in the first B rows of its own buffers
        const bool pl = pool_last && l + 1 == c.layers;
        const int Tn = pl ? c.B : T, Rn = pl ? c.H : c.L * c.H;
        const int32_t* cun = pl ? nullptr : A.cu;
        const float* x_in = a.x_in;
        const bf16_t* ctx1 = a.ctx1;
        if (pl) {
            SPN_TRYF(gather_pool_rows(a.x_in, a.ctx1, A.zero_idx, packed ? A.cu : nullptr, c.L, A.pool_x, A.pool_ctx, c.B, W, st));
            x_in = A.pool_x; ctx1 = A.pool_ctx;
        }
        SPN_TRYF(nt_resid(ctx1, Bw(BO_SA_WO), Tn, W, W, P(LO_SA_BO), x_in, a.y1, st));
        SPN_TRYF(layernorm_fwd(a.y1, P(LO_SA_LNG), P(LO_SA_LNB), a.x1b, a.x1, a.mean1, a.rstd1, Tn, W, 1e-12f, st));
        // cross-attention over the image tokens
        SPN_TRYF(nt(a.x1b, Bw(BO_CA_WQ), Tn, W, W, P(LO_CA_BQ), a.q2, nullptr, st));
        if (absorb) {
            // scores = (0.125 q_h Wk_h) X^T, ctx_h = (softmax X) Wv_h^T + bv_h: K and V are never formed (xattn.hip)
            SPN_TRYF(xattn_head_expand(a.q2, W, Bw(BO_CA_WKV_T), 2 * W, 0, a.qa, Tn, c.H, E, 0.125f, st));
            SPN_TRYF(xattn_scores_softmax(a.qa, A.enc_b, a.pm, c.B, Rn, c.S, E, st, cun, c.H));
            SPN_TRYF(xattn_apply(a.pm, A.enc_b, a.oa, c.B, Rn, c.S, E, st, cun, c.H, (int64_t)Tn * c.H));
            SPN_TRYF(xattn_head_contract(a.oa, Bw(BO_CA_WKV), W, P(LO_CA_BKV), a.ctx2, W, Tn, c.H, E, 1.0f, st));
        } else {
            SPN_TRYF(nt(A.enc_b, Bw(BO_CA_WKV), TS, 2 * W, E, P(LO_CA_BKV), a.kv2, nullptr, st));
            SPN_TRYF(attention_fwd(cross_attn_args(c, a), st));
        }
        SPN_TRYF(nt_resid(a.ctx2, Bw(BO_CA_WO), Tn, W, W, P(LO_CA_BO), a.x1, a.y2, st));
        SPN_TRYF(layernorm_fwd(a.y2, P(LO_CA_LNG), P(LO_CA_LNB), a.x2b, a.x2, a.mean2, a.rstd2, Tn, W, 1e-12f, st));
        // feed-forward (BertIntermediate exact GELU + BertOutput)
        SPN_TRYF(nt(a.x2b, Bw(BO_FF_W1), Tn, I, W, P(LO_FF_B1), a.u, nullptr, st, ACT_GELU_ERF, a.pre));
        SPN_TRYF(nt_resid(a.u, Bw(BO_FF_W2), Tn, W, I, P(LO_FF_B2), a.x2, a.y3, st));
        SPN_TRYF(layernorm_fwd(a.y3, P(LO_FF_LNG), P(LO_FF_LNB), xb_out, x_out, a.mean3, a.rstd3, Tn, W, 1e-12f, st));
    }
    // text_proj of the [ENC] position (blip_cir.py:98); the L2-normalise is spn_combine_l2norm_fwd
    // pooled last layer: the first B rows of x_final ARE the [ENC] rows
    if (!pool_last) {
        if (packed) SPN_TRYF(gather_rows_abs(A.x_final, A.cu, A.h0, c.B, W, st));
        else SPN_TRYF(gather_rows_f32(A.x_final, A.zero_idx, A.h0, c.B, c.L, W, st));
    }
    SPN_TRYF(cast_f32_bf16(pool_last ? A.x_final : A.h0, A.h0b, (size_t)c.B * W, st));
    SPN_TRYF(nt(A.h0b, wb + t.bf16_proj, c.B, c.Dp, W, params + t.proj_b, nullptr, proj_out, st));
    return SPN_OK;
}

// Deferred weight gradients (as in the text tower, tower.hip): each layer keeps the dY operands of its seven products
// dW = dY^T X in buffers of its own - dyb after each of the three LayerNorm backward passes [T,W] x3, dpre [T,I],
// dq of the cross-attention [T,W], dqkv of the self-attention [T,3W], dkv of the cross-attention [B*S,2W] - and the
// products of ALL layers run behind the data path as grouped launches without split-K (gemm_tn_grouped): the six with
// the text rows as reduction (36 problems per launch), and the cross-attention K/V projection of all layers
// (reduction over the B*S image tokens).  SPN_TN_GROUP=0: one split-K launch per product as before.
static size_t fusion_defer_layer_bytes(const FusionCfg& c) {
    const size_t T = (size_t)fusion_rows(c), TS = (size_t)c.B * c.S, W = c.W, I = c.I;
    // the K/V-projection operand: dkv2 [B*S, 2W], or - absorbed form - dQ' [T, H, E] and the cross-attention's dctx [T, W]
    const size_t kvop = fusion_absorb(c) ? fa(fusion_the(c) * 2) + fa(T * W * 2) : fa(TS * 2 * W * 2);
    return 4 * fa(T * W * 2) + fa(T * I * 2) + fa(T * 3 * W * 2) + kvop +
           3 * fa(layernorm_bwd_workspace_bytes((int)T, (int)W));     // + row partials of the three LayerNorm backward passes
}

static bool fusion_tn_group_on() {
    static const bool on = [] {
        const char* e = spn_env("SPN_TN_GROUP");
        return !(e && e[0] == '0');
    }();
    return on;
}

size_t fusion_ws_bytes(const FusionCfg& c) {
    const size_t T = (size_t)fusion_rows(c), TS = (size_t)c.B * c.S, W = c.W, I = c.I;
    size_t b = 0;
    b += fa(T * W * 4) + fa(T * W * 4) + fa(T * W * 2);          // dx, dy, dyb
    b += fa(T * I * 2);                                          // dpre
    b += 2 * fa(T * W * 2);                                      // dctx, dctx2 (pooled last layer: scatter target)
    b += fa(T * 3 * W * 2);                                      // dqkv (also dq2)
    if (fusion_absorb(c)) b += 2 * fa(fusion_the(c) * 2) + fa(fusion_rsp(c) * 2);   // dO', dQ', dS
    else b += fa(TS * 2 * W * 2);                                // dkv2
    b += fa((size_t)c.B * c.H * c.L * 4);                        // delta
    b += fa((size_t)c.B * c.Dp * 2) + fa((size_t)c.B * W * 4);   // dproj bf16, dh0
    // per-layer dY operands of the deferred weight gradients; only when that path is on (SPN_TN_GROUP != 0 and few
    // enough layers for the grouped launches - the same test as spn_fusion_bwd's)
    const bool grouped = fusion_tn_group_on() && c.layers * 6 <= 2 * TN_GROUP_MAX && c.layers <= TN_GROUP_MAX;
    b += fusion_defer_layer_bytes(c) * (grouped ? c.layers : 0);
    size_t op = 0;
    auto mx = [&](size_t v) { if (v > op) op = v; };
    mx(gemm_tn_workspace_bytes((int)T, (int)W, (int)I));
    mx(gemm_tn_workspace_bytes((int)T, (int)I, (int)W));
    mx(gemm_tn_workspace_bytes((int)T, (int)W, (int)W));
    mx(gemm_tn_workspace_bytes((int)T, 3 * (int)W, (int)W));
    mx(gemm_tn_workspace_bytes((int)TS, 2 * (int)W, c.E));
    mx(gemm_tn_workspace_bytes(c.B, c.Dp, (int)W));
    mx(gemm_tn_grouped_workspace_bytes((int)TS));
    mx(layernorm_bwd_workspace_bytes((int)T, (int)W));
    mx(embed_bwd_packed_ws_bytes(c.L, c.W));              // packed embedding backward: reserved in the dense size too (see fusion_act_bytes)
    return b + fa(op);
}

// dproj: gradient w.r.t. the text_proj output [B, Dp] (i.e. after spn_combine_l2norm_bwd)
// phases (bit mask): 1 = head (text_proj), 2 = the layers [l_lo, l_hi) from the top down incl. their deferred weight
// gradients (final when the call returns), 4 = tail (embeddings).  The residual gradient lives in the workspace between the
// calls (the carve-up is identical in every phase), so a data-parallel caller can hand a group's gradient range to its
// all-reduce while the layers below are still running (spn_fusion_bwd_phase; FusionEncoder.backward_phased).
static int fusion_bwd_impl(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
                           const float* dproj, float* grads, char* ws, size_t ws_bytes, int phases, int l_lo, int l_hi,
                           hipStream_t st);

int fusion_bwd(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
               const float* dproj, float* grads, char* ws, size_t ws_bytes, hipStream_t st) {
    return fusion_bwd_impl(c, params, wb, ids, acts, dproj, grads, ws, ws_bytes, 7, 0, c.layers, st);
}

int fusion_bwd_phase(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
                     const float* dproj, float* grads, char* ws, size_t ws_bytes, int phase, int l_lo, int l_hi, hipStream_t st) {
    if (phase < 0 || phase > 2) return SPN_ERR_ARG;
    if (phase == 1 && (l_lo < 0 || l_hi > c.layers || l_lo >= l_hi)) return SPN_ERR_ARG;
    return fusion_bwd_impl(c, params, wb, ids, acts, dproj, grads, ws, ws_bytes, 1 << phase, l_lo, l_hi, st);
}

static int fusion_bwd_impl(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
                           const float* dproj, float* grads, char* ws, size_t ws_bytes, int phases, int l_lo, int l_hi,
                           hipStream_t st) {
    SPN_TRYF(fusion_check(c));
    if (ws_bytes < fusion_ws_bytes(c)) return SPN_ERR_WORKSPACE;
    FusionLayout t;
    fusion_layout(c, &t);
    FusionActs A = fusion_acts_at(acts, c);
    const int T = fusion_rows(c), TS = c.B * c.S, W = c.W, I = c.I, E = c.E;
    const size_t Ts = (size_t)T;
    char* p = ws;
    auto take = [&](size_t bytes) { char* r = p; p += fa(bytes); return r; };
    float* dx = (float*)take(Ts * W * 4);        // gradient w.r.t. the current LN output
    float* dy = (float*)take(Ts * W * 4);        // gradient w.r.t. the current LN input
    bf16_t* dyb = (bf16_t*)take(Ts * W * 2);
    bf16_t* dpre = (bf16_t*)take(Ts * I * 2);
    bf16_t* dctx = (bf16_t*)take(Ts * W * 2);
    bf16_t* dctx2 = (bf16_t*)take(Ts * W * 2);
    bf16_t* dqkv = (bf16_t*)take(Ts * 3 * W * 2);
    const bool absorb = fusion_absorb(c), pool_last = fusion_pool_last(c);
    bf16_t *dkv2 = nullptr, *doa = nullptr, *dqa = nullptr, *dsm = nullptr;
    if (absorb) {
        doa = (bf16_t*)take(fusion_the(c) * 2); dqa = (bf16_t*)take(fusion_the(c) * 2); dsm = (bf16_t*)take(fusion_rsp(c) * 2);
    } else {
        dkv2 = (bf16_t*)take((size_t)TS * 2 * W * 2);
    }
    float* delta = (float*)take((size_t)c.B * c.H * c.L * 4);
    bf16_t* dprojb = (bf16_t*)take((size_t)c.B * c.Dp * 2);
    float* dh0 = (float*)take((size_t)c.B * W * 4);
    char* defer_base = p;
    const size_t defer_stride = fusion_defer_layer_bytes(c);
    const bool grouped = fusion_tn_group_on() && c.layers * 6 <= 2 * TN_GROUP_MAX && c.layers <= TN_GROUP_MAX;
    p += defer_stride * (grouped ? c.layers : 0);
    TnProblem qT[2 * TN_GROUP_MAX], qS[TN_GROUP_MAX];
    int nT = 0, nS = 0;
    const size_t lnp = fa(layernorm_bwd_workspace_bytes(T, W));
    FoldBatch fb{};                                   // LayerNorm parameter gradients: folded in batches behind the loop
    fb.n = layernorm_bwd_partial_rows(T);
    fb.stride = (size_t)2 * W;
    fb.C = (size_t)2 * W;
    auto fold_flush = [&]() -> int {
        if (!fb.items) return SPN_OK;
        const int rc = fold_rows_batched(fb, st);
        fb.items = 0;
        return rc;
    };
    float* opws = (float*)p;
    const size_t opws_bytes = ws_bytes - (size_t)(p - ws);

    // head: text_proj
    if (phases & 1) {
        if (!dproj) return SPN_ERR_ARG;
        SPN_TRYF(cast_f32_bf16(dproj, dprojb, (size_t)c.B * c.Dp, st));
        SPN_TRYF(gemm_tn(dprojb, A.h0b, c.B, c.Dp, W, c.Dp, W, grads + t.proj_w, W, 1.0f, 0, grads + t.proj_b, opws,
                         opws_bytes, st));
        // pooled last layer: its output rows ARE the B [ENC] rows - the gradient goes straight into the first B rows of dx
        SPN_TRYF(nt(dprojb, wb + t.bf16_proj_t, c.B, W, c.Dp, nullptr, nullptr, pool_last ? dx : dh0, st));
        if (pool_last) {}
        else if (c.T > 0) SPN_TRYF(scatter_rows_abs(dh0, A.row_b, A.cu, dx, nullptr, T, W, st));
        else SPN_TRYF(scatter_rows_f32(dh0, A.zero_idx, dx, nullptr, c.B, c.L, W, st));
    }

    for (int l = (phases & 2) ? l_hi - 1 : -1; l >= l_lo; --l) {
        FusionLayerActs a = fusion_layer_acts_at(A.layers + A.layer_bytes * l, c);
        const float* pp = params + t.layers + t.layer_size * l;
        float* gp = grads + t.layers + t.layer_size * l;
        const bf16_t* b = wb + t.bf16_layer_size * l;
        auto P = [&](int i) { return pp + t.layer_off[i]; };
        auto G = [&](int i) { return gp + t.layer_off[i]; };
        auto Bw = [&](int i) { return b + t.bf16_off[i]; };
        // the dY operands of this layer's weight gradients: buffers of its own when they are deferred
        bf16_t *dyb_ff = dyb, *dyb_ca = dyb, *dyb_sa = dyb, *dpre_l = dpre, *dq_ca = dqkv, *dqkv_sa = dqkv, *dkv2_l = dkv2;
        bf16_t *dqa_l = dqa, *dctx_ca = dctx;
        float* ln_part = nullptr;
        // pooled last layer: everything above its self-attention on the B [ENC] rows (first B rows of the buffers), its weight
        // and LayerNorm gradients finished on the spot (their reductions run over B rows, not T)
        const bool pl = pool_last && l + 1 == c.layers;
        const int Tn = pl ? c.B : T;
        if (grouped) {
            char* q = defer_base + defer_stride * l;
            auto tk = [&](size_t bytes) { char* r = q; q += fa(bytes); return (bf16_t*)r; };
            dyb_ff = tk(Ts * W * 2); dyb_ca = tk(Ts * W * 2); dyb_sa = tk(Ts * W * 2); dq_ca = tk(Ts * W * 2);
            dpre_l = tk(Ts * I * 2); dqkv_sa = tk(Ts * 3 * W * 2);
            if (absorb) { dqa_l = tk(fusion_the(c) * 2); dctx_ca = tk(Ts * W * 2); }
            else dkv2_l = tk((size_t)TS * 2 * W * 2);
            ln_part = (float*)q;
        }
        // LayerNorm backward: grouped mode leaves the [dgamma | dbeta] row partials in the layer's buffer (they are
        // adjacent in the parameter layout) and queues their fold
        auto ln_bwd = [&](const float* xin, int lo_g, int lo_b, const float* mean, const float* rstd, bf16_t* out_b, int k) -> int {
            if (grouped && !pl && G(lo_b) == G(lo_g) + W) {
                float* part = (float*)((char*)ln_part + lnp * k);
                SPN_TRYF(layernorm_bwd(nullptr, dx, xin, P(lo_g), mean, rstd, dy, 0, out_b, G(lo_g), G(lo_b), 2, T, W, part, lnp, st));
                fb.ws[fb.items] = part; fb.out[fb.items++] = G(lo_g);
                if (fb.items == FOLD_BATCH_MAX) SPN_TRYF(fold_flush());
                return SPN_OK;
            }
            return layernorm_bwd(nullptr, dx, xin, P(lo_g), mean, rstd, dy, 0, out_b, G(lo_g), G(lo_b), 0, Tn, W, opws, opws_bytes, st);
        };
        auto wgrad = [&](const bf16_t* Aop, const bf16_t* Bop, int Kr, int N1, int N2, float* Cw, float* cb) -> int {
            if (!grouped || (Kr != T && Kr != TS))
                return gemm_tn(Aop, Bop, Kr, N1, N2, N1, N2, Cw, N2, 1.0f, 0, cb, opws, opws_bytes, st);
            TnProblem pr{Aop, Bop, Cw, cb, N1, N2, N1, N2, N2};
            if (Kr == T) qT[nT++] = pr;
            else qS[nS++] = pr;
            return SPN_OK;
        };
        // ---- FFN: x3 = LN(y3), y3 = x2 + gelu(x2 W1^T + b1) W2^T + b2
        SPN_TRYF(ln_bwd(a.y3, LO_FF_LNG, LO_FF_LNB, a.mean3, a.rstd3, dyb_ff, 0));
        {
            GemmEpilogue e;
            e.aux_in = a.pre; e.aux_grad = 1; e.act = ACT_GELU_ERF; e.out_bf16 = dpre_l; e.ldc = I;
            SPN_TRYF(gemm_nt(dyb_ff, Bw(BO_FF_W2_T), Tn, I, W, W, W, GEMM_DACT, e, st));
        }
        SPN_TRYF(wgrad(dyb_ff, a.u, Tn, W, I, G(LO_FF_W2), G(LO_FF_B2)));
        SPN_TRYF(wgrad(dpre_l, a.x2b, Tn, I, W, G(LO_FF_W1), G(LO_FF_B1)));
        SPN_TRYF(nt_resid(dpre_l, Bw(BO_FF_W1_T), Tn, W, I, nullptr, dy, dx, st));        // dx = d/dx2
        // ---- cross-attention: x2 = LN(y2), y2 = x1 + attn(q(x1), kv(enc)) Wo^T + bo
        SPN_TRYF(ln_bwd(a.y2, LO_CA_LNG, LO_CA_LNB, a.mean2, a.rstd2, dyb_ca, 1));
        SPN_TRYF(nt(dyb_ca, Bw(BO_CA_WO_T), Tn, W, W, nullptr, dctx_ca, nullptr, st));
        SPN_TRYF(wgrad(dyb_ca, a.ctx2, Tn, W, W, G(LO_CA_WO), G(LO_CA_BO)));
        if (absorb) {
            const int R = pl ? c.H : c.L * c.H;
            const int32_t* cun = pl ? nullptr : A.cu;
            SPN_TRYF(xattn_delta(dctx_ca, a.ctx2, P(LO_CA_BKV) + W, delta, Tn, c.H, st));
            SPN_TRYF(xattn_head_expand(dctx_ca, W, Bw(BO_CA_WKV_T), 2 * W, W, doa, Tn, c.H, E, 1.0f, st));         // dO' = dctx_h Wv_h
            SPN_TRYF(xattn_dscores(doa, A.enc_b, a.pm, delta, dsm, c.B, R, c.S, E, st, cun, c.H));                // dS
            SPN_TRYF(xattn_apply(dsm, A.enc_b, dqa_l, c.B, R, c.S, E, st, cun, c.H, (int64_t)Tn * c.H));                            // dQ' = dS X
            SPN_TRYF(xattn_head_contract(dqa_l, Bw(BO_CA_WKV), 0, nullptr, dq_ca, W, Tn, c.H, E, 0.125f, st));    // dq_h
            if (!grouped || pl)
                SPN_TRYF(xattn_wgrad(a.q2, 0, dqa_l, 0, dctx_ca, 0, a.oa, 0, G(LO_CA_WKV), G(LO_CA_BKV), 0, 1, Tn, W, c.H, E, 0.125f, st));
        } else {
            AttnBwdArgs g;
            g.f = cross_attn_args(c, a);
            g.d_o = dctx_ca; g.lddo = W;
            g.dq = dq_ca; g.lddq = W;
            g.dk = dkv2_l; g.dv = dkv2_l + W; g.lddk = g.lddv = 2 * W;
            g.delta = delta;
            SPN_TRYF(attention_bwd(g, st));
            SPN_TRYF(wgrad(dkv2_l, A.enc_b, TS, 2 * W, E, G(LO_CA_WKV), G(LO_CA_BKV)));
        }
        SPN_TRYF(wgrad(dq_ca, a.x1b, Tn, W, W, G(LO_CA_WQ), G(LO_CA_BQ)));
        SPN_TRYF(nt_resid(dq_ca, Bw(BO_CA_WQ_T), Tn, W, W, nullptr, dy, dx, st));         // dx = d/dx1
        // ---- self-attention: x1 = LN(y1), y1 = x_in + attn(qkv(x_in)) Wo^T + bo
        SPN_TRYF(ln_bwd(a.y1, LO_SA_LNG, LO_SA_LNB, a.mean1, a.rstd1, dyb_sa, 2));
        SPN_TRYF(nt(dyb_sa, Bw(BO_SA_WO_T), Tn, W, W, nullptr, dctx, nullptr, st));
        SPN_TRYF(wgrad(dyb_sa, pl ? A.pool_ctx : a.ctx1, Tn, W, W, G(LO_SA_WO), G(LO_SA_BO)));
        const bf16_t* dctx_sa = dctx;
        const float* dres = dy;                                  // gradient of the residual stream at the layer input
        if (pl) {
            // back to all rows: d/dy1 [B, W] (dy) -> dx at the [ENC] rows, d/dctx1 [B, W] -> dctx2 at the [ENC] rows, zeros elsewhere
            SPN_TRYF(scatter_pool_rows(dy, dctx, A.zero_idx, c.T > 0 ? A.row_b : nullptr, A.cu, c.L, dx, dctx2, T, W, st));
            dctx_sa = dctx2; dres = dx;
        }
        {
            AttnBwdArgs g;
            g.f = self_attn_args(c, a, A.key_bias, A.cu);
            g.d_o = dctx_sa; g.lddo = W;
            g.dq = dqkv_sa; g.dk = dqkv_sa + W; g.dv = dqkv_sa + 2 * W;
            g.lddq = g.lddk = g.lddv = 3 * W;
            g.delta = delta;
            SPN_TRYF(attention_bwd(g, st));
        }
        SPN_TRYF(wgrad(dqkv_sa, a.xb_in, T, 3 * W, W, G(LO_SA_WQKV), G(LO_SA_BQKV)));
        SPN_TRYF(nt_resid(dqkv_sa, Bw(BO_SA_WQKV_T), T, W, 3 * W, nullptr, dres, dx, st));  // dx = d/dx_in
    }
    // the deferred weight gradients: everything with the text rows as reduction (<= 48 problems per launch), then the
    // cross-attention K/V projections of all layers (reduction over the image tokens)
    for (int i = 0; i < nT; i += TN_GROUP_MAX)
        SPN_TRYF(gemm_tn_grouped(qT + i, nT - i < TN_GROUP_MAX ? nT - i : TN_GROUP_MAX, T, opws, opws_bytes, st));
    if (nS) SPN_TRYF(gemm_tn_grouped(qS, nS, TS, opws, opws_bytes, st));
    const int l_hi_def = (pool_last && l_hi == c.layers) ? l_hi - 1 : l_hi;      // the pooled last layer finished its own
    if (absorb && grouped && (phases & 2) && l_hi_def > l_lo) {
        // absorbed K/V weight gradients of the layers [l_lo, l_hi) in one launch: the operands sit at constant strides
        FusionLayerActs a0 = fusion_layer_acts_at(A.layers + A.layer_bytes * l_lo, c);
        char* q = defer_base + defer_stride * l_lo;
        q += 4 * fa(Ts * W * 2) + fa(Ts * I * 2) + fa(Ts * 3 * W * 2);          // the carve-up of the loop above
        const bf16_t* dqa0 = (const bf16_t*)q;
        const bf16_t* dctx0 = (const bf16_t*)(q + fa(fusion_the(c) * 2));
        float* g0 = grads + t.layers + t.layer_size * l_lo;
        SPN_TRYF(xattn_wgrad(a0.q2, A.layer_bytes / 2, dqa0, defer_stride / 2, dctx0, defer_stride / 2, a0.oa, A.layer_bytes / 2,
                             g0 + t.layer_off[LO_CA_WKV], g0 + t.layer_off[LO_CA_BKV], (size_t)t.layer_size, l_hi_def - l_lo, T, W, c.H,
                             E, 0.125f, st));
    }
    SPN_TRYF(fold_flush());
    if (!(phases & 4)) return SPN_OK;
    // embeddings: x0 = LN(word[ids] + pos)
    SPN_TRYF(layernorm_bwd(nullptr, dx, A.emb, params + t.emb_ln_g, A.emb_mean, A.emb_rstd, dy, 0, nullptr, grads + t.emb_ln_g,
                           grads + t.emb_ln_b, 0, T, W, opws, opws_bytes, st));
    SPN_TRYF(zero_fill_f32(grads + t.word, (size_t)c.vocab * W, st));
    if (c.T > 0)
        SPN_TRYF(embed_bwd_packed(ids, A.row_b, A.row_l, A.cu, dy, grads + t.word, grads + t.pos, T, c.B, c.L, W, c.vocab, opws,
                                  opws_bytes, st));
    else SPN_TRYF(embed_bwd(ids, A.last, dy, grads + t.word, grads + t.pos, c.B, c.L, W, c.vocab, st));
    if (c.L < c.max_pos) {
        SPN_TRYF(zero_fill_f32(grads + t.pos + (size_t)c.L * W, (size_t)(c.max_pos - c.L) * W, st));
    }
    return SPN_OK;
}

}  // namespace spn
