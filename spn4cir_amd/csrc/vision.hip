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
    t->n_bf16 = t->bf16_proj_t + (int64_t)c.W * c.D;
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
    SPN_TRYV(cast_transpose_f32_bf16(params + t.proj, nullptr, wb + t.bf16_proj_t, c.W, c.D, st));
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
    if (tokens_out)   // vit.py:195: x = self.norm(x) over every token
        SPN_TRYV(layernorm_fwd(cur, params + t.ln_post_g, params + t.ln_post_b, nullptr, tokens_out, nullptr, nullptr, (int)T,
                               c.W, bc.eps, st));
    GemmEpilogue e;
    e.out_f32 = feats; e.ldc = c.D;
    if (c.kind == 1) e.bias = params + t.proj_b;
    SPN_TRYV(gemm_nt(ln_cls, wb + t.bf16_proj_t, c.B, c.D, c.W, c.W, c.W, GEMM_STORE, e, st));
    (void)g;
    return SPN_OK;
}

}  // namespace spn
