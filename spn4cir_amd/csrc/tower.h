// Tower-level structs (internal C++; the C-ABI mirrors of TextCfg / TextLayout live in
// include/spn4cir_hip.h and must stay layout-compatible).
#pragma once
#include "kernels.h"

namespace spn {

struct BlockCfg {
    int B, L, W, H, causal, act;
    float eps;
    // packed (variable-length) rows: T live rows in total, sequence b = rows cu[b]..cu[b+1]-1 (device pointer);
    // T == 0 means the dense B*L layout
    int T = 0;
    const int32_t* cu = nullptr;
    int fuse_resid = 0;   // block_fwd: residual adds deferred into the following LayerNorm (tower.hip)
    int rows() const { return T > 0 ? T : B * L; }
};

struct BlockParams {
    const float *ln1_g, *ln1_b, *b_qkv, *b_o, *ln2_g, *ln2_b, *b_fc, *b_proj;
    const bf16_t *w_qkv, *w_qkv_t, *w_o, *w_o_t, *w_fc, *w_fc_t, *w_proj, *w_proj_t;
};

struct BlockGrads {
    float *ln1_g, *ln1_b, *w_qkv, *b_qkv, *w_o, *b_o, *ln2_g, *ln2_b, *w_fc, *b_fc, *w_proj, *b_proj;
};

struct BlockActs {
    float* x_in;
    float *mean1, *rstd1;
    bf16_t* h1;
    bf16_t* qkv;
    float* lse;
    bf16_t* attn;
    float* x_mid;
    float *mean2, *rstd2;
    bf16_t* h2;
    bf16_t* pre;      // act'(pre-activation) of the MLP (GemmEpilogue::aux_grad): what the backward multiplies by
    bf16_t* u;
    float* x_out;
};

struct TextCfg {       // == spn_text_cfg
    int B, L, L_ctx, W, H, layers, D, vocab;
    int T;             // packed live rows (sum of the sequence lengths), 0 = dense B*L rows
    int pool;          // != 0: the last block runs its out-projection / MLP on the B pooled (EOT) rows only (tower.hip)
};

struct TextLayout {    // == spn_text_layout_t (element offsets)
    int64_t tok, pos, blocks, block_size, lnf_g, lnf_b, text_proj, n_params;
    int64_t block_off[13];
    int64_t bf16_block_size, bf16_text_proj, bf16_text_proj_t, n_bf16;
};

struct VisionCfg {     // == spn_vision_cfg
    int B, res, patch, W, H, layers, D;
    int kind;          // 0 = CLIP VisionTransformer, 1 = BLIP / timm ViT (blip4cir/vit.py)
};

struct VisionLayout {  // == spn_vision_layout_t (element offsets)
    int64_t conv1, conv_b, cls, pos, ln_pre_g, ln_pre_b, blocks, block_size, ln_post_g, ln_post_b, proj, proj_b, n_params;
    int64_t block_off[13];
    int64_t bf16_conv1, bf16_blocks, bf16_block_size, bf16_proj_t, n_bf16, kp, seq;
    int64_t bf16_proj;   // proj [W, D] as stored (backward-data operand); bf16_proj_t is its transpose
};

void vision_layout(const VisionCfg& c, VisionLayout* t);
size_t vision_ws_bytes(const VisionCfg& c);
int vision_refresh_bf16(const VisionCfg& c, const float* params, bf16_t* wb, hipStream_t st);
int vision_fwd(const VisionCfg& c, const float* params, const bf16_t* wb, const float* image, char* ws, size_t ws_bytes,
               float* feats, float* tokens_out, hipStream_t st);

struct FusionCfg {     // == spn_fusion_cfg
    int B, L, S, W, H, layers, I, E, Dp, vocab, max_pos;
    int T;             // packed live text rows (sum of the caption lengths), 0 = dense B*L rows
};

struct FusionLayout {  // == spn_fusion_layout_t (element offsets)
    int64_t word, pos, emb_ln_g, emb_ln_b, layers, layer_size, proj_w, proj_b, n_params;
    int64_t layer_off[21];
    int64_t bf16_layer_size, bf16_proj, bf16_proj_t, n_bf16;
    int64_t bf16_off[15];
};

void fusion_layout(const FusionCfg& c, FusionLayout* t);
int fusion_packed_ok(const FusionCfg& c);
size_t fusion_act_bytes(const FusionCfg& c);
size_t fusion_ws_bytes(const FusionCfg& c);
int fusion_refresh_bf16(const FusionCfg& c, const float* params, bf16_t* wb, hipStream_t st);
int fusion_fwd(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, const int32_t* mask,
               const float* enc, char* acts, float* proj_out, hipStream_t st, const bf16_t* token_bank = nullptr,
               const int64_t* token_idx = nullptr, int64_t bank_rows = 0);
int fusion_bwd(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
               const float* dproj, float* grads, char* ws, size_t ws_bytes, hipStream_t st);
// phase 0 = head (needs dproj), 1 = layers [l_lo, l_hi) top-down with their weight gradients, 2 = tail (embeddings)
int fusion_bwd_phase(const FusionCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
                     const float* dproj, float* grads, char* ws, size_t ws_bytes, int phase, int l_lo, int l_hi, hipStream_t st);

void block_param_offsets(int W, int64_t off[13]);
int64_t block_bf16_size(int W);
BlockParams block_params_at(const float* p, const bf16_t* wb, int W);
BlockGrads block_grads_at(float* g, int W);
int block_refresh_bf16(const float* p, bf16_t* wb, int W, hipStream_t st);
int blocks_refresh_bf16(const float* p, int64_t p_stride, bf16_t* wb, int64_t wb_stride, int layers, int W, hipStream_t st);
size_t block_act_bytes(const BlockCfg& c);
BlockActs block_acts_at(char* base, const BlockCfg& c);
size_t block_bwd_scratch_bytes(const BlockCfg& c);
size_t block_op_ws_bytes(const BlockCfg& c);
int block_fwd(const BlockCfg& c, const BlockParams& P, const BlockActs& A, hipStream_t st, const float* x_prev = nullptr,
              bf16_t* y_out = nullptr);
// Optional overlap of the weight-gradient (TN) GEMMs with the rest of a block's backward: they feed nothing
// downstream, so they run on a side stream behind events while the main stream continues with the data path
// (NT GEMMs, LayerNorm / attention backward).  dxb_alt: second bf16 residual-gradient buffer [T, W] (the TN
// GEMMs still read the old one while LayerNorm backward writes the new one); ws2: their own split-K workspace.
struct BwdOverlap {
    hipStream_t side;
    hipEvent_t ev[6];
    bf16_t* dxb_alt;
    float* ws2;
    size_t ws2_bytes;
};
// Cross-block deferral of the weight gradients: the block keeps the four dY operands in buffers of its own (instead of
// the shared scratch / the in-place bf16 residual gradient), writes the four products it owes into `problems` and
// launches nothing; the caller batches the problems of several blocks into one gemm_tn_grouped launch.
struct BwdDefer {
    bf16_t* dpre;        // [T, 4W]
    bf16_t* dqkv;        // [T, 3W]
    bf16_t* dx_mid;      // [T, W]  residual gradient between the two halves of the block
    bf16_t* dx_out;      // [T, W]  bf16 gradient leaving the block (dx_bf16, the one entering it, stays intact)
    float* ln_partials;  // 2 x layernorm_bwd_workspace_bytes: [ln_2 | ln_1] row partials of dgamma / dbeta, folded by the caller
    TnProblem* problems; // [4]
};
// dxb_group (optional, [T, W] bf16): enables the grouped weight-gradient launch at the end of the block (tower.hip)
int block_bwd(const BlockCfg& c, const BlockParams& P, const BlockActs& A, const BlockGrads& G, float* dx,
              bf16_t* dx_bf16, char* scratch, float* ws, size_t ws_bytes, hipStream_t st, const BwdOverlap* ov = nullptr,
              bf16_t* dxb_group = nullptr, const BwdDefer* defer = nullptr);

// training path (CLIP kind 0 only): activations kept per layer, clip4cir/models.py:156-158 (wo_bank first stage)
size_t vision_train_act_bytes(const VisionCfg& c);
size_t vision_bwd_ws_bytes(const VisionCfg& c);
int vision_fwd_train(const VisionCfg& c, const float* params, const bf16_t* wb, const float* image, char* acts, float* feats,
                     hipStream_t st);
int vision_bwd(const VisionCfg& c, const float* params, const bf16_t* wb, char* acts, const float* dfeats, float* grads,
               char* ws, size_t ws_bytes, hipStream_t st);

// exact.hip: fp32-exact inference of the CLIP towers (f32-input MFMA GEMMs, fp32 attention) for validation
int gemm_f32(const float* A, const float* B, int M, int N, int K, int lda, int ldb, int b_kn, const float* bias, int act,
             const float* resid, int ldr, float* C, int ldc, float alpha, hipStream_t st);
int im2col3x3_f32(const float* x, float* out, int B, int H, int W, int C, int stride, int nchw, int ldk, hipStream_t st);
int avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int k, hipStream_t st);
// bf16 fast path (resnet.hip)
int im2col3x3_nhwc_bf16(const bf16_t* x, bf16_t* out, int B, int H, int W, int Cp, int stride, hipStream_t st);
int im2col3x3_stem_bf16(const float* image, bf16_t* out, int B, int H, int W, int stride, hipStream_t st);
int relu_add_bf16(bf16_t* y, const bf16_t* resid, size_t n, hipStream_t st);
int avgpool_nhwc_bf16(const bf16_t* x, bf16_t* y, int B, int H, int W, int Cp, int k, hipStream_t st);
int attnpool_tokens_f32(const float* x, const float* pos, float* tok, int B, int HW, int C, hipStream_t st);
int attnpool_attend_f32(const float* q, const float* k, const float* v, float* out, int B, int S, int H, hipStream_t st);
size_t text_exact_ws_bytes(const TextCfg& c);
int text_fwd_exact(const TextCfg& c, const float* params, const int32_t* ids, char* ws, size_t ws_bytes, float* feats,
                   hipStream_t st);
size_t vision_exact_ws_bytes(const VisionCfg& c);
int vision_fwd_exact(const VisionCfg& c, const float* params, const float* image, char* ws, size_t ws_bytes, float* feats,
                     hipStream_t st);

void text_layout(const TextCfg& c, TextLayout* t);
size_t text_act_bytes(const TextCfg& c);
size_t text_ws_bytes(const TextCfg& c);
int text_refresh_bf16(const TextCfg& c, const float* params, bf16_t* wb, hipStream_t st);
// cu_seqlens: device [B+1] prefix sums of the live lengths (EOT position + 1), required iff c.T > 0
int text_fwd(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, const int32_t* cu_seqlens,
             char* acts, float* feats, hipStream_t st);
int text_bwd(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
             const float* dfeats, float* grads, char* ws, size_t ws_bytes, hipStream_t st);
int text_fwd_tokens(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts, float* feats,
                    float* tokens, bf16_t* tokens_bf16, float* tok_mean, float* tok_rstd, hipStream_t st);
int text_bwd_tokens(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
                    const float* dfeats, const float* dtokens, const float* tok_mean, const float* tok_rstd, float* grads,
                    char* ws, size_t ws_bytes, hipStream_t st);
int text_bwd_tokens_head(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, const float* dfeats,
                         const float* dtokens, const float* tok_mean, const float* tok_rstd, float* grads, char* ws,
                         size_t ws_bytes, hipStream_t st);
int text_bwd_tail_tokens(const TextCfg& c, const int32_t* ids, char* acts, float* grads, char* ws, size_t ws_bytes,
                         hipStream_t st);
int text_bwd_head(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, const float* dfeats,
                  float* grads, char* ws, size_t ws_bytes, hipStream_t st);
int text_bwd_layer(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, float* grads, int l, char* ws,
                   size_t ws_bytes, hipStream_t st);
// Deferred variant: everything of block l except its four weight-gradient GEMMs (their operands stay in per-layer
// buffers of the workspace); text_bwd_wgrad computes them for the blocks [l_begin, l_end) in one grouped launch - call
// it after the deferred backward of all of them (at most TN_GROUP_MAX / 4 blocks per call).
int text_bwd_layer_deferred(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, float* grads, int l,
                            char* ws, size_t ws_bytes, hipStream_t st);
int text_bwd_wgrad(const TextCfg& c, char* acts, float* grads, int l_begin, int l_end, char* ws, size_t ws_bytes,
                   hipStream_t st);
int text_bwd_tail(const TextCfg& c, const int32_t* ids, char* acts, float* grads, char* ws, size_t ws_bytes,
                  hipStream_t st);

}  // namespace spn
