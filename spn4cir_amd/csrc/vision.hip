// CLIP VisionTransformer forward (clip4cir/clip/model.py:206-242), inference only: the image tower
// is frozen in stage 2 (models_negplus.py:27-28) and runs for bank extraction and validation
// (models_negplus.py:59-125, utils.py:24-50).  Patch embedding = im2col + the NT GEMM (conv1 has
// stride == kernel and no bias), then class token + positional embedding + ln_pre, the same
// residual blocks as the text tower without a mask, ln_post on the class token and `x @ proj`.
#include "tower.h"

namespace spn {

#define SPN_TRYV(x)                       \
    do {                                  \
        int rc__ = (x);                   \
        if (rc__ != SPN_OK) return rc__;  \
    } while (0)

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// patches[b*g*g + gy*g + gx][c*p*p + ky*p + kx] = image[b][c][gy*p+ky][gx*p+kx]; columns >= 3pp are zero
__global__ void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int R, int p, int Kp) {
    const int g = R / p, K = 3 * p * p;
    const size_t total = (size_t)B * g * g * Kp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kp);
        const size_t row = i / Kp;
        float v = 0.f;
        if (col < K) {
            const int c = col / (p * p), ky = (col / p) % p, kx = col % p;
            const int gx = (int)(row % g), gy = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
            v = img[(((size_t)b * 3 + c) * R + gy * p + ky) * R + gx * p + kx];
        }
        out[i] = f2bf(v);
    }
}

// x[b,0,:] = cls + pos[0]; x[b,1+i,:] = emb[b*g*g+i,:] + pos[1+i]      (model.py:227-230)
__global__ void assemble_tokens_kernel(const float* __restrict__ emb, const float* __restrict__ cls,
                                       const float* __restrict__ pos, float* __restrict__ x, int B, int S, int W) {
    const int w4 = W >> 2;
    const size_t total = (size_t)B * S * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % w4) * 4;
        const size_t row = i / w4;
        const int s = (int)(row % S), b = (int)(row / S);
        f32x4 v = s == 0 ? *(const f32x4*)(cls + c) : *(const f32x4*)(emb + ((size_t)b * (S - 1) + s - 1) * W + c);
        v += *(const f32x4*)(pos + (size_t)s * W + c);
        *(f32x4*)(x + row * W + c) = v;
    }
}

static int grid1d(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

static BlockCfg vision_block_cfg(const VisionCfg& c) {
    BlockCfg b;
    const int g = c.res / c.patch;
    b.B = c.B; b.L = g * g + 1; b.W = c.W; b.H = c.H; b.causal = 0;
    b.act = c.kind == 1 ? ACT_GELU_ERF : ACT_QUICKGELU;
    b.eps = c.kind == 1 ? 1e-6f : 1e-5f;
    return b;
}

static int vision_kp(const VisionCfg& c) { return ((3 * c.patch * c.patch + 63) / 64) * 64; }

void vision_layout(const VisionCfg& c, VisionLayout* t) {
    int64_t bo[13];
    block_param_offsets(c.W, bo);
    const int g = c.res / c.patch, S = g * g + 1;
    const int64_t K = 3ll * c.patch * c.patch;
    int64_t o = 0;
    t->conv1 = o; o += (int64_t)c.W * K;
    t->conv_b = o; o += c.W;          // used by kind 1 only (CLIP's conv1 has no bias)
    t->cls = o; o += c.W;
    t->pos = o; o += (int64_t)S * c.W;
    t->ln_pre_g = o; o += c.W;
    t->ln_pre_b = o; o += c.W;
    t->blocks = o; t->block_size = bo[12]; o += bo[12] * c.layers;
    t->ln_post_g = o; o += c.W;
    t->ln_post_b = o; o += c.W;
    t->proj = o; o += (int64_t)c.W * c.D;
    t->proj_b = o; o += c.D;          // kind 1: vision_proj bias
    t->n_params = o;
    for (int i = 0; i < 13; ++i) t->block_off[i] = bo[i];
    t->bf16_conv1 = 0;
    t->bf16_blocks = (int64_t)c.W * vision_kp(c);
    t->bf16_block_size = block_bf16_size(c.W);
    t->bf16_proj_t = t->bf16_blocks + t->bf16_block_size * c.layers;
    t->bf16_proj = t->bf16_proj_t + (int64_t)c.W * c.D;
    t->n_bf16 = t->bf16_proj + (int64_t)c.W * c.D;
    t->kp = vision_kp(c);
    t->seq = S;
}

static int vision_check(const VisionCfg& c) {
    if (c.B <= 0 || c.layers <= 0 || c.patch <= 0 || c.res % c.patch) return SPN_ERR_ARG;
    if (c.W % 64 || c.H * 64 != c.W || c.D % 4) return SPN_ERR_SHAPE;
    return SPN_OK;
}

// conv1 [W, 3pp] -> bf16 [W, Kp] zero padded: pad on the host side of the cast via a strided kernel
__global__ void pad_cast_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int rows, int K, int Kp) {
    const size_t total = (size_t)rows * Kp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kp);
        const size_t r = i / Kp;
        y[i] = f2bf(col < K ? x[r * K + col] : 0.f);
    }
}

int vision_refresh_bf16(const VisionCfg& c, const float* params, bf16_t* wb, hipStream_t st) {
    SPN_TRYV(vision_check(c));
    VisionLayout t;
    vision_layout(c, &t);
    const int K = 3 * c.patch * c.patch;
    hipLaunchKernelGGL(pad_cast_rows_kernel, dim3(grid1d((size_t)c.W * t.kp)), dim3(256), 0, st, params + t.conv1,
                       wb + t.bf16_conv1, c.W, K, (int)t.kp);
    SPN_CHECK_LAUNCH();
    for (int l = 0; l < c.layers; ++l)
        SPN_TRYV(block_refresh_bf16(params + t.blocks + t.block_size * l, wb + t.bf16_blocks + t.bf16_block_size * l, c.W, st));
    SPN_TRYV(cast_transpose_f32_bf16(params + t.proj, wb + t.bf16_proj, wb + t.bf16_proj_t, c.W, c.D, st));
    return SPN_OK;
}

size_t vision_ws_bytes(const VisionCfg& c) {
    const BlockCfg bc = vision_block_cfg(c);
    const size_t T = (size_t)c.B * bc.L, P = (size_t)c.B * (bc.L - 1);
    size_t b = 0;
    b += al256(P * vision_kp(c) * 2);     // patches
    b += al256(P * c.W * 4);              // patch embeddings
    b += al256(T * c.W * 4);              // assembled tokens
    b += block_act_bytes(bc);             // one block's activations, reused by every layer
    b += al256(T * c.W * 4);              // ping-pong residual stream
    b += al256((size_t)c.B * 4);          // zero row indices (class token)
    b += al256((size_t)c.B * c.W * 4);    // class rows
    b += al256((size_t)c.B * c.W * 2);    // ln_post output
    return b;
}

int vision_fwd(const VisionCfg& c, const float* params, const bf16_t* wb, const float* image, char* ws, size_t ws_bytes,
               float* feats, float* tokens_out, hipStream_t st) {
    SPN_TRYV(vision_check(c));
    if (ws_bytes < vision_ws_bytes(c)) return SPN_ERR_WORKSPACE;
    VisionLayout t;
    vision_layout(c, &t);
    const BlockCfg bc = vision_block_cfg(c);
    const int S = bc.L, g = c.res / c.patch, Kp = (int)t.kp;
    const size_t T = (size_t)c.B * S, P = (size_t)c.B * (S - 1);
    char* p = ws;
    auto take = [&](size_t bytes) { char* r = p; p += al256(bytes); return r; };
    bf16_t* patches = (bf16_t*)take(P * Kp * 2);
    float* emb = (float*)take(P * c.W * 4);
    float* tok = (float*)take(T * c.W * 4);
    char* acts = p; p += block_act_bytes(bc);
    float* xb = (float*)take(T * c.W * 4);
    int32_t* zero_idx = (int32_t*)take((size_t)c.B * 4);
    float* cls_rows = (float*)take((size_t)c.B * c.W * 4);
    bf16_t* ln_cls = (bf16_t*)take((size_t)c.B * c.W * 2);

    hipLaunchKernelGGL(im2col_kernel, dim3(grid1d(P * Kp)), dim3(256), 0, st, image, patches, c.B, c.res, c.patch, Kp);
    SPN_CHECK_LAUNCH();
    {
        GemmEpilogue e;
        e.out_f32 = emb; e.ldc = c.W;
        if (c.kind == 1) e.bias = params + t.conv_b;       // timm PatchEmbed conv has a bias
        SPN_TRYV(gemm_nt(patches, wb + t.bf16_conv1, (int)P, c.W, Kp, Kp, Kp, GEMM_STORE, e, st));
    }
    BlockActs A = block_acts_at(acts, bc);
    float* xa = A.x_in;      // first block input: ln_pre(tokens) for CLIP, the tokens themselves for the timm ViT
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(grid1d(T * (c.W / 4))), dim3(256), 0, st, emb, params + t.cls,
                       params + t.pos, c.kind == 1 ? xa : tok, c.B, S, c.W);
    SPN_CHECK_LAUNCH();
    if (c.kind != 1)
        SPN_TRYV(layernorm_fwd(tok, params + t.ln_pre_g, params + t.ln_pre_b, nullptr, xa, nullptr, nullptr, (int)T, c.W,
                               1e-5f, st));
    float* cur = xa;
    float* nxt = xb;
    for (int l = 0; l < c.layers; ++l) {
        BlockActs a = A;
        a.x_in = cur;
        a.x_out = nxt;
        a.pre = nullptr;     // inference: the pre-activation copy is not needed
        const BlockParams Pm = block_params_at(params + t.blocks + t.block_size * l,
                                               wb + t.bf16_blocks + t.bf16_block_size * l, c.W);
        SPN_TRYV(block_fwd(bc, Pm, a, st));
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    hipError_t he = hipMemsetAsync(zero_idx, 0, (size_t)c.B * 4, st);
    if (he != hipSuccess) return (int)he;
    SPN_TRYV(gather_rows_f32(cur, zero_idx, cls_rows, c.B, S, c.W, st));
    SPN_TRYV(layernorm_fwd(cls_rows, params + t.ln_post_g, params + t.ln_post_b, ln_cls, nullptr, nullptr, nullptr, c.B, c.W,
                           bc.eps, st));
    if (tokens_out && c.kind == 1)   // vit.py:195: x = self.norm(x) over every token
        SPN_TRYV(layernorm_fwd(cur, params + t.ln_post_g, params + t.ln_post_b, nullptr, tokens_out, nullptr, nullptr, (int)T,
                               c.W, bc.eps, st));
    else if (tokens_out) {           // CLIP: the transformer output itself (ln_post only touches the class token,
        // clip/model.py:237); TG-CIR's extract_img_fea (tgcir/models.py:99-123) consumes these rows
        he = hipMemcpyAsync(tokens_out, cur, T * c.W * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (he != hipSuccess) return (int)he;
    }
    GemmEpilogue e;
    e.out_f32 = feats; e.ldc = c.D;
    if (c.kind == 1) e.bias = params + t.proj_b;
    SPN_TRYV(gemm_nt(ln_cls, wb + t.bf16_proj_t, c.B, c.D, c.W, c.W, c.W, GEMM_STORE, e, st));
    (void)g;
    return SPN_OK;
}

// ------------------------------------------------------------------------------ training
// First-stage / in-batch configuration (clip4cir/models.py:31-33,151-167: the visual tower is trainable when
// wo_bank).  The reference recomputes the tower under torch.utils.checkpoint; with 288 GB of HBM the per-layer
// activations are simply kept (ViT-L/14 at 256 images: ~58 GB).
struct VisionActs {
    bf16_t* patches;
    float* emb;
    float* tok;
    float *mean_pre, *rstd_pre;
    char* blocks;
    size_t block_bytes;
    float* x_final;
    int32_t* zero_idx;
    float* cls_rows;
    float *mean_f, *rstd_f;
    bf16_t* ln_cls;
};

size_t vision_train_act_bytes(const VisionCfg& c) {
    const BlockCfg bc = vision_block_cfg(c);
    const size_t T = (size_t)c.B * bc.L, P = (size_t)c.B * (bc.L - 1);
    size_t b = al256(P * vision_kp(c) * 2) + al256(P * c.W * 4) + al256(T * c.W * 4) + 2 * al256(T * 4);
    b += block_act_bytes(bc) * c.layers + al256(T * c.W * 4);
    b += al256((size_t)c.B * 4) + al256((size_t)c.B * c.W * 4) + 2 * al256((size_t)c.B * 4) + al256((size_t)c.B * c.W * 2);
    return b;
}

static VisionActs vision_acts_at(char* base, const VisionCfg& c) {
    const BlockCfg bc = vision_block_cfg(c);
    const size_t T = (size_t)c.B * bc.L, P = (size_t)c.B * (bc.L - 1);
    VisionActs A;
    char* p = base;
    auto take = [&](size_t bytes) { char* r = p; p += al256(bytes); return r; };
    A.patches = (bf16_t*)take(P * vision_kp(c) * 2);
    A.emb = (float*)take(P * c.W * 4);
    A.tok = (float*)take(T * c.W * 4);
    A.mean_pre = (float*)take(T * 4);
    A.rstd_pre = (float*)take(T * 4);
    A.block_bytes = block_act_bytes(bc);
    A.blocks = p; p += A.block_bytes * c.layers;
    A.x_final = (float*)take(T * c.W * 4);
    A.zero_idx = (int32_t*)take((size_t)c.B * 4);
    A.cls_rows = (float*)take((size_t)c.B * c.W * 4);
    A.mean_f = (float*)take((size_t)c.B * 4);
    A.rstd_f = (float*)take((size_t)c.B * 4);
    A.ln_cls = (bf16_t*)take((size_t)c.B * c.W * 2);
    return A;
}

int vision_fwd_train(const VisionCfg& c, const float* params, const bf16_t* wb, const float* image, char* acts, float* feats,
                     hipStream_t st) {
    SPN_TRYV(vision_check(c));
    if (c.kind != 0) return SPN_ERR_ARG;     // only the CLIP tower is ever trained on this path
    VisionLayout t;
    vision_layout(c, &t);
    const BlockCfg bc = vision_block_cfg(c);
    const int S = bc.L, Kp = (int)t.kp;
    const size_t T = (size_t)c.B * S, P = (size_t)c.B * (S - 1);
    VisionActs A = vision_acts_at(acts, c);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid1d(P * Kp)), dim3(256), 0, st, image, A.patches, c.B, c.res, c.patch, Kp);
    SPN_CHECK_LAUNCH();
    {
        GemmEpilogue e;
        e.out_f32 = A.emb; e.ldc = c.W;
        SPN_TRYV(gemm_nt(A.patches, wb + t.bf16_conv1, (int)P, c.W, Kp, Kp, Kp, GEMM_STORE, e, st));
    }
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3(grid1d(T * (c.W / 4))), dim3(256), 0, st, A.emb, params + t.cls,
                       params + t.pos, A.tok, c.B, S, c.W);
    SPN_CHECK_LAUNCH();
    BlockActs first = block_acts_at(A.blocks, bc);
    SPN_TRYV(layernorm_fwd(A.tok, params + t.ln_pre_g, params + t.ln_pre_b, nullptr, first.x_in, A.mean_pre, A.rstd_pre, (int)T,
                           c.W, 1e-5f, st));
    for (int l = 0; l < c.layers; ++l) {
        BlockActs a = block_acts_at(A.blocks + A.block_bytes * l, bc);
        a.x_out = (l + 1 < c.layers) ? block_acts_at(A.blocks + A.block_bytes * (l + 1), bc).x_in : A.x_final;
        const BlockParams Pm = block_params_at(params + t.blocks + t.block_size * l,
                                               wb + t.bf16_blocks + t.bf16_block_size * l, c.W);
        SPN_TRYV(block_fwd(bc, Pm, a, st));
    }
    hipError_t he = hipMemsetAsync(A.zero_idx, 0, (size_t)c.B * 4, st);
    if (he != hipSuccess) return (int)he;
    SPN_TRYV(gather_rows_f32(A.x_final, A.zero_idx, A.cls_rows, c.B, S, c.W, st));
    SPN_TRYV(layernorm_fwd(A.cls_rows, params + t.ln_post_g, params + t.ln_post_b, A.ln_cls, nullptr, A.mean_f, A.rstd_f, c.B,
                           c.W, bc.eps, st));
    GemmEpilogue e;
    e.out_f32 = feats; e.ldc = c.D;
    SPN_TRYV(gemm_nt(A.ln_cls, wb + t.bf16_proj_t, c.B, c.D, c.W, c.W, c.W, GEMM_STORE, e, st));
    return SPN_OK;
}

// demb[b*(S-1)+i, :] = bf16(dtok[b*S+1+i, :])
__global__ void patch_rows_bf16_kernel(const float* __restrict__ dtok, bf16_t* __restrict__ demb, int B, int S, int W) {
    const int w4 = W >> 2;
    const size_t total = (size_t)B * (S - 1) * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % w4) * 4;
        const size_t row = i / w4;
        const size_t b = row / (S - 1), s = row % (S - 1);
        const f32x4 v = *(const f32x4*)(dtok + (b * S + 1 + s) * W + c);
        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)(demb + row * W + c) = o;
    }
}

__global__ void unpad_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int K, int Kp) {
    const size_t total = (size_t)rows * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        y[i] = x[(i / K) * Kp + i % K];
}

struct VisionBwdWs {
    char* scratch;
    float* dx;
    bf16_t* dxb;
    bf16_t* dfb;
    float *dln, *dcls, *dtok, *dconv;
    bf16_t* demb;
    float* opws;
    size_t opws_bytes;
};

static size_t vision_bwd_fixed_bytes(const VisionCfg& c) {
    const BlockCfg bc = vision_block_cfg(c);
    const size_t T = (size_t)c.B * bc.L, P = (size_t)c.B * (bc.L - 1);
    size_t b = block_bwd_scratch_bytes(bc);
    b += al256(T * c.W * 4) + al256(T * c.W * 2) + al256((size_t)c.B * c.D * 2) + 2 * al256((size_t)c.B * c.W * 4);
    b += al256(T * c.W * 4) + al256((size_t)c.W * vision_kp(c) * 4) + al256(P * c.W * 2);
    return b;
}

size_t vision_bwd_ws_bytes(const VisionCfg& c) {
    const BlockCfg bc = vision_block_cfg(c);
    const int T = c.B * bc.L, P = c.B * (bc.L - 1);
    size_t op = block_op_ws_bytes(bc);
    auto mx = [&](size_t v) { if (al256(v) > op) op = al256(v); };
    mx(gemm_tn_workspace_bytes(c.B, c.W, c.D));
    mx(gemm_tn_workspace_bytes(P, c.W, vision_kp(c)));
    mx(layernorm_bwd_workspace_bytes(T, c.W));
    mx(layernorm_bwd_workspace_bytes(c.B, c.W));
    return vision_bwd_fixed_bytes(c) + op;
}

int vision_bwd(const VisionCfg& c, const float* params, const bf16_t* wb, char* acts, const float* dfeats, float* grads,
               char* ws, size_t ws_bytes, hipStream_t st) {
    SPN_TRYV(vision_check(c));
    if (c.kind != 0) return SPN_ERR_ARG;
    if (c.D % 8) return SPN_ERR_SHAPE;
    if (ws_bytes < vision_bwd_ws_bytes(c)) return SPN_ERR_WORKSPACE;
    VisionLayout t;
    vision_layout(c, &t);
    const BlockCfg bc = vision_block_cfg(c);
    const int S = bc.L, Kp = (int)t.kp, K = 3 * c.patch * c.patch;
    const size_t T = (size_t)c.B * S, P = (size_t)c.B * (S - 1);
    VisionActs A = vision_acts_at(acts, c);
    VisionBwdWs w;
    {
        char* p = ws;
        auto take = [&](size_t bytes) { char* r = p; p += al256(bytes); return r; };
        w.scratch = p; p += block_bwd_scratch_bytes(bc);
        w.dx = (float*)take(T * c.W * 4);
        w.dxb = (bf16_t*)take(T * c.W * 2);
        w.dfb = (bf16_t*)take((size_t)c.B * c.D * 2);
        w.dln = (float*)take((size_t)c.B * c.W * 4);
        w.dcls = (float*)take((size_t)c.B * c.W * 4);
        w.dtok = (float*)take(T * c.W * 4);
        w.dconv = (float*)take((size_t)c.W * Kp * 4);
        w.demb = (bf16_t*)take(P * c.W * 2);
        w.opws = (float*)p;
        w.opws_bytes = ws_bytes - (size_t)(p - ws);
    }
    // x[:, 0] -> ln_post -> @ proj      (clip/model.py:238-241)
    SPN_TRYV(cast_f32_bf16(dfeats, w.dfb, (size_t)c.B * c.D, st));
    SPN_TRYV(gemm_tn(A.ln_cls, w.dfb, c.B, c.W, c.D, c.W, c.D, grads + t.proj, c.D, 1.0f, 0, nullptr, w.opws, w.opws_bytes, st));
    {
        GemmEpilogue e;
        e.out_f32 = w.dln; e.ldc = c.W;
        SPN_TRYV(gemm_nt(w.dfb, wb + t.bf16_proj, c.B, c.W, c.D, c.D, c.D, GEMM_STORE, e, st));
    }
    SPN_TRYV(layernorm_bwd(nullptr, w.dln, A.cls_rows, params + t.ln_post_g, A.mean_f, A.rstd_f, w.dcls, 0, nullptr,
                           grads + t.ln_post_g, grads + t.ln_post_b, 0, c.B, c.W, w.opws, w.opws_bytes, st));
    SPN_TRYV(scatter_rows_f32(w.dcls, A.zero_idx, w.dx, w.dxb, c.B, S, c.W, st));
    for (int l = c.layers - 1; l >= 0; --l) {
        BlockActs a = block_acts_at(A.blocks + A.block_bytes * l, bc);
        const BlockParams Pm = block_params_at(params + t.blocks + t.block_size * l,
                                               wb + t.bf16_blocks + t.bf16_block_size * l, c.W);
        const BlockGrads G = block_grads_at(grads + t.blocks + t.block_size * l, c.W);
        SPN_TRYV(block_bwd(bc, Pm, a, G, w.dx, w.dxb, w.scratch, w.opws, w.opws_bytes, st));
    }
    // ln_pre, positional / class embedding, conv1 (model.py:224-231)
    SPN_TRYV(layernorm_bwd(nullptr, w.dx, A.tok, params + t.ln_pre_g, A.mean_pre, A.rstd_pre, w.dtok, 0, nullptr,
                           grads + t.ln_pre_g, grads + t.ln_pre_b, 0, (int)T, c.W, w.opws, w.opws_bytes, st));
    SPN_TRYV(embed_bwd(nullptr, nullptr, w.dtok, nullptr, grads + t.pos, c.B, S, c.W, 0, st));
    hipError_t he = hipMemcpyAsync(grads + t.cls, grads + t.pos, (size_t)c.W * 4, hipMemcpyDeviceToDevice, st);
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(grads + t.conv_b, 0, (size_t)c.W * 4, st);     // unused by CLIP (no conv bias)
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(grads + t.proj_b, 0, (size_t)c.D * 4, st);
    if (he != hipSuccess) return (int)he;
    hipLaunchKernelGGL(patch_rows_bf16_kernel, dim3(grid1d(P * (c.W / 4))), dim3(256), 0, st, w.dtok, w.demb, c.B, S, c.W);
    SPN_CHECK_LAUNCH();
    SPN_TRYV(gemm_tn(w.demb, A.patches, (int)P, c.W, Kp, c.W, Kp, w.dconv, Kp, 1.0f, 0, nullptr, w.opws, w.opws_bytes, st));
    hipLaunchKernelGGL(unpad_rows_kernel, dim3(grid1d((size_t)c.W * K)), dim3(256), 0, st, w.dconv, grads + t.conv1, c.W, K, Kp);
    SPN_CHECK_LAUNCH();
    return SPN_OK;
}

}  // namespace spn
