// Internal C++ interface between the kernel translation units and the C-ABI (api.hip).
#pragma once
#include "common.h"
#include <stddef.h>

namespace spn {

enum { GEMM_STORE = 0, GEMM_RESID = 1, GEMM_DACT = 2, GEMM_BANKSTATS = 3 };
enum { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_GELU_ERF = 2, ACT_RELU_POST = 3 /* gemm_f32 only: ReLU after bias + residual */ };

struct GemmEpilogue {
    const float* bias = nullptr;     // [N] fp32, added before everything else
    const float* resid = nullptr;    // GEMM_RESID: fp32 [M, ldr] added to the result
    const bf16_t* aux_in = nullptr;  // GEMM_DACT: pre-activation, result *= act'(aux_in)
    bf16_t* aux_out = nullptr;       // GEMM_STORE with act: pre-activation copy (bf16)
    int aux_grad = 0;                // 1: aux_out receives act'(pre-activation) instead, and GEMM_DACT multiplies by
                                     // aux_in as it is - the derivative shares the forward's sigmoid / erf, so the
                                     // backward epilogue loses its transcendentals (the training towers use this)
    bf16_t* out_bf16 = nullptr;
    float* out_f32 = nullptr;
    int ldc = 0;                     // leading dim of out_bf16 / out_f32 / aux_in / aux_out
    int ldr = 0;
    int act = ACT_NONE;
    float alpha = 1.0f;
    int direct_store = 0;            // gemm2 only: 1 = store from the MFMA layout (no LDS staging)
    int stag_from = 0, stag_to = 0, stag_ticks = 0;   // gemm2: first-round workgroups [stag_from, stag_to) start stag_ticks (100 MHz) late
    int col_group = 0;               // gemm2: > 0 = tile order in column groups of this many 256-wide tiles (launch_nt2)
    int aux_ld = 0;                  // gemm2, full tiles, GEMM_DACT: cache-policy bits of the aux_in loads (0 plain, 2 nt, 16 sc1, 17 sc0 sc1)
    int store_wt = 0;                // gemm2, full tiles: 1 = outputs leave through write-through (sc1) buffer stores - the output
                                     // stream then does not displace the operand panels from the XCD's L2 (launch_nt2)
    // GEMM_BANKSTATS (gemm2, 256x256 tile only): A = queries [B, D], B = bank rows [M, D]; nothing is stored but the
    // per-row softmax statistics of each 256-column tile: bs_out[(tile_n * B + row) * 4] = {max, sum exp, sum, label logit}
    // of logits * bs_inv_tau (the layout bank_stats_fold_kernel reduces)
    const int64_t* bs_labels = nullptr;
    float* bs_out = nullptr;
    float bs_inv_tau = 1.0f;
    int bs_m_begin = 0;
    // optional: keep the tile-relative probabilities p = exp(logit - tile max) as bf16 [B, bs_ldp] (bs_ldp a multiple of 256)
    // and the tile maxima [tile_n][B] for the GEMM-path backward pass (bank_grad_q with saved logits, B >= 128)
    bf16_t* bs_p_out = nullptr;
    int bs_ldp = 0;
    float* bs_max_out = nullptr;
    int dbg = 0;                     // gemm2 only, SPN_GEMM_DBG bottleneck-elimination bits (wrong results): 1 = every k tile
                                     // re-reads k0 = 0, 2 = no DMA after the prologue, 4 = LDS fragments read once per
                                     // k step, 8 = no epilogue
};

// gemm.hip
int gemm_nt(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode,
            const GemmEpilogue& ep, hipStream_t st);
// colsum_out (optional, [N1]) receives the column sums of A (the bias gradient when A = dY)
int gemm_tn(const bf16_t* A, const bf16_t* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
            float alpha, int accumulate, float* colsum_out, float* ws, size_t ws_bytes, hipStream_t st);
size_t gemm_tn_workspace_bytes(int Kr, int N1, int N2);
// gemm2.hip (second generation; gemm_nt / gemm_tn dispatch to these unless SPN_GEMM_V1=1)
int gemm_nt2(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode,
             const GemmEpilogue& ep, hipStream_t st);
int gemm_nt2_mid(const bf16_t* A, const bf16_t* B, int M, int N, int K, int lda, int ldb, int mode, const GemmEpilogue& ep,
                 hipStream_t st, int variant);
// logits statistics of q [B, D] against bank [M, D] on the GEMM path; partial [ceil(M/256)][B][4] floats
int gemm_bank_stats(const bf16_t* q, const bf16_t* bank, int B, int M, int D, int ldq, int ldb, const int64_t* labels,
                    float inv_tau, int m_begin, float* partial, hipStream_t st, bf16_t* p_out = nullptr, int ldp = 0,
                    float* max_out = nullptr);
int gemm_bank_stats_tiles(int M, int D);
int gemm_bank_stats_bn(int D);      // bank rows per statistics tile (128 for D <= 256, else 256)
bool gemm_tn2_pair_ok(int N1a, int N2a);
size_t gemm_tn2_pair_workspace_bytes(int Kr, int N1a, int N2a, int N1b, int N2b);
int gemm_tn2_pair(const bf16_t* A1, const bf16_t* B1, int N1a, int N2a, int lda1, int ldb1, float* C1, int ldc1, float* cs1,
                  const bf16_t* A2, const bf16_t* B2, int N1b, int N2b, int lda2, int ldb2, float* C2, int ldc2, float* cs2,
                  int Kr, float* ws, size_t ws_bytes, hipStream_t st);
int gemm_tn2(const bf16_t* A, const bf16_t* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
             float alpha, int accumulate, float* colsum_out, float* ws, size_t ws_bytes, hipStream_t st);
// Grouped weight gradients (gemm2.hip): C_p [N1, N2] = A_p [Kr, N1]^T B_p [Kr, N2] (overwritten, fp32), colsum_p [N1] =
// column sums of A_p (optional), for up to TN_GROUP_MAX problems sharing Kr, in one launch without split-K
// (only the tiles of a last, mostly empty round are split; ws holds their slabs).
struct TnProblem {
    const bf16_t* A; const bf16_t* B;
    float* C; float* colsum;
    int N1, N2, lda, ldb, ldc;
};
constexpr int TN_GROUP_MAX = 48;
size_t gemm_tn_grouped_workspace_bytes(int Kr);
int device_cu_count();   // compute units of the current device (256 on MI355X)
int gemm_tn_grouped(const TnProblem* probs, int n, int Kr, float* ws, size_t ws_bytes, hipStream_t st);
size_t gemm_tn2_workspace_bytes(int Kr, int N1, int N2);
int gemm_nt_persist_set(int v);    // gemm2.hip: -1 = environment default, 0 / 1 = persistent multi-round NT kernel off / on (experiments build)
bool gemm_use_v1();
int gemm_cfg();

// negtype.hip: the four in-batch InfoNCE terms of clip4cir/models_negtype.py (forward + feature gradients, fp32)
size_t negtype_workspace_bytes(int B, int D);
int negtype_head(const float* R, const float* T, const float* I, int B, int D, float inv_tau, int neg_type, float* loss,
                 float* dR, float* dT, float* dI, float* ws, size_t ws_bytes, hipStream_t st);

// elementwise.hip
int cast_f32_bf16(const float* x, bf16_t* y, size_t n, hipStream_t st);
int cast_bf16_f32(const bf16_t* x, float* y, size_t n, hipStream_t st);
int sum_ranks_bf16(const bf16_t* x, int G, size_t m, bf16_t* out, hipStream_t st);
int sum_ranks_f32(const float* x, int G, size_t m, float* out, hipStream_t st);
int gather_bank_rows_bf16(const bf16_t* bank, const int64_t* idx, int64_t n_rows, bf16_t* out, int B, size_t row_elems, hipStream_t st);
int tau_grad(const float* q, const float* dqk, int lddq, const float* tau, int B, int D, float alpha, const float* scale_dev,
             float* dtau, float* inv_tau, hipStream_t st);
// y[b, :D] = bf16(x[b, :] * s), y[b, D:ldo] = 0; s = *scale_dev (device scalar) or its reciprocal
int scale_cast_bf16(const float* x, const float* scale_dev, int reciprocal, bf16_t* y, int B, int D, int ldo, hipStream_t st);
int transpose_bf16(const bf16_t* x, bf16_t* y, int rows, int cols, hipStream_t st);             // y[c][r] = x[r][c]
int cast_transpose_f32_bf16(const float* x, bf16_t* y, bf16_t* yt, int rows, int cols, hipStream_t st);
struct CastTransposeSet {           // up to 4 matrices per repetition (element offsets), see cast_transpose_multi
    int64_t src[4], dst[4], dst_t[4];
    int rows[4], cols[4];
    int tile_start[5];              // prefix sums of the 32x32 tile counts; [4] = tiles per repetition
    int64_t p_stride, wb_stride;
};
int cast_transpose_multi(const float* p, bf16_t* wb, const CastTransposeSet& d, int layers, hipStream_t st);
int colsum_bf16(const bf16_t* x, int rows, int cols, int ld, float* out, int accumulate, float* ws, size_t ws_bytes,
                hipStream_t st);
size_t colsum_workspace_bytes(int rows, int cols);
int fold_rows(const float* ws, size_t stride, int n, size_t C, float* out, float alpha, int accumulate, hipStream_t st);
int zero_fill_f32(float* p, size_t n, hipStream_t st);
// jpeg.hip: baseline JPEG batch decode (Huffman -> IDCT -> upsample + colour), see include/spn4cir_hip.h spn_jpeg_decode_batch
int jpeg_decode_batch(const uint8_t* bytes, const void* images, int n_images, const void* segs, int n_segs, const void* huff,
                      const uint16_t* qtabs, int16_t* coefs, size_t coef_elems, uint8_t* planes, uint8_t* rgb, int max_blocks,
                      int max_pixels, hipStream_t st);     // elementwise.hip: the library's own zero fill (no hipMemsetAsync on the step)
constexpr int FOLD_BATCH_MAX = 32;
struct FoldBatch {            // out[i][c] = sum_{r < n} ws[i][r * stride + c], c < C, for i < items
    const float* ws[FOLD_BATCH_MAX];
    float* out[FOLD_BATCH_MAX];
    int items, n;
    size_t stride, C;
};
int fold_rows_batched(const FoldBatch& b, hipStream_t st);
int layernorm_bwd_partial_rows(int rows);     // rows of the [.][2W] partial matrix layernorm_bwd leaves in ws
int embed_fwd(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* x, int B, int L, int W, int vocab,
              hipStream_t st);
int embed_bwd(const int32_t* ids, const int32_t* eot, const float* dx, float* dtok, float* dpos, int B, int L, int W,
              int vocab, hipStream_t st);
size_t embed_bwd_all_ws_bytes(int B, int L, int W);
int embed_bwd_all(const int32_t* ids, const float* dx, float* dtok, float* dpos, int B, int L, int W, int vocab, int hot_id,
                  float* ws, size_t ws_bytes, hipStream_t st);
int eot_argmax(const int32_t* ids, int32_t* eot, int B, int L, hipStream_t st);
int gather_rows_f32(const float* x, const int32_t* eot, float* out, int B, int L, int W, hipStream_t st);
int scatter_rows_f32(const float* src, const int32_t* eot, float* dx, bf16_t* dx_bf16, int B, int L, int W,
                     hipStream_t st);
int build_row_map(const int32_t* cu, int32_t* row_b, int32_t* row_l, int32_t* eot_row, int B, hipStream_t st, int32_t* cu_copy = nullptr);
int embed_fwd_packed(const int32_t* ids, const int32_t* row_b, const int32_t* row_l, const float* tok_emb,
                     const float* pos_emb, float* x, int T, int L, int W, int vocab, hipStream_t st);
size_t embed_bwd_packed_ws_bytes(int L, int W);
int embed_bwd_packed(const int32_t* ids, const int32_t* row_b, const int32_t* row_l, const int32_t* cu, const float* dx,
                     float* dtok, float* dpos, int T, int B, int L, int W, int vocab, float* ws, size_t ws_bytes,
                     hipStream_t st);
int gather_rows_abs(const float* x, const int32_t* rows, float* out, int B, int W, hipStream_t st);
// pooled rows of the text tower's last block: row(b) = rows_abs ? rows_abs[b] : b * L + eot[b]
int gather_pool_rows(const float* x, const bf16_t* a, const int32_t* eot, const int32_t* rows_abs, int L, float* xo, bf16_t* ao,
                     int B, int W, hipStream_t st);
int scatter_pool_rows(const float* de, const bf16_t* da, const int32_t* eot, const int32_t* row_b, const int32_t* eot_row, int L,
                      float* dx, bf16_t* dattn, int T, int W, hipStream_t st);
int scatter_rows_abs(const float* src, const int32_t* row_b, const int32_t* eot_row, float* dx, bf16_t* dx_bf16, int T,
                     int W, hipStream_t st);
int adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
               float wd, int step, float inv_scale, const float* found_inf, hipStream_t st, const float* grad_scale = nullptr,
               const float* step_dev = nullptr);
int adamw_tick(float* step_dev, const float* found_inf, hipStream_t st);
int grad_unscale_check(float* g, size_t n, float inv_scale, float* found_inf, hipStream_t st);

// norm.hip
int layernorm_fwd(const float* x, const float* gamma, const float* beta, bf16_t* y_bf16, float* y_f32, float* mean,
                  float* rstd, int rows, int W, float eps, hipStream_t st);
int layernorm_fwd_add(const float* x, const bf16_t* y, const float* gamma, const float* beta, float* x_out, bf16_t* y_bf16,
                      float* mean, float* rstd, int rows, int W, float eps, hipStream_t st);
int layernorm_bwd(const bf16_t* dy_bf16, const float* dy_f32, const float* x, const float* gamma, const float* mean,
                  const float* rstd, float* dx, int accumulate_dx, bf16_t* dx_bf16, float* dgamma, float* dbeta,
                  int accumulate_dparam, int rows, int W, float* ws, size_t ws_bytes, hipStream_t st);
size_t layernorm_bwd_workspace_bytes(int rows, int W);

// attention.hip
struct AttnArgs {
    const bf16_t *q, *k, *v;         // row b*Lq+i (q) / b*Lk+j (k,v); head h at column h*64
    int ldq, ldk, ldv;
    bf16_t* o; int ldo;              // [B*Lq, H*64]
    float* lse;                      // [B, H, Lq] log-sum-exp of the scaled scores
    const float* key_bias;           // optional additive bias per key [B, Lk] (BERT padding mask), may be null
    int B, H, Lq, Lk, causal;
    float scale;
    // packed (variable-length) self-attention: sequence b occupies rows cu[b]..cu[b+1]-1 of q/k/v/o (Lq = Lk =
    // the maximum length, used for the kernel geometry and the lse layout); nullptr = dense b*Lq rows
    const int32_t* cu = nullptr;
};
struct AttnBwdArgs {
    AttnArgs f;
    const bf16_t* d_o; int lddo;
    bf16_t *dq, *dk, *dv; int lddq, lddk, lddv;
    float* delta;                    // workspace [B, H, Lq]
};
int attention_fwd(const AttnArgs& a, hipStream_t st);
int attention_bwd(const AttnBwdArgs& a, hipStream_t st);

// xattn.hip: cross-attention over frozen image tokens with the K/V projections absorbed into the query / output side
// (rows of a sample r = l*H + h; R = L*H; SP = xattn_sp(S) columns in P / dS)
bool xattn_absorb_ok(int B, int L, int H, int S, int E, int W);
int xattn_sp(int S);
int xattn_head_expand(const bf16_t* A, int lda, const bf16_t* Wt, int ldw, int col0, bf16_t* out, int T, int H, int E, float alpha,
                      hipStream_t st);
int xattn_head_contract(const bf16_t* A, const bf16_t* Wr, int row0, const float* bias, bf16_t* out, int ldo, int T, int H, int E,
                        float alpha, hipStream_t st);
// cu != null (packed text rows): sample b owns the rows cu[b]*H .. cu[b+1]*H; R = the longest sample's row count (grid only)
int xattn_scores_softmax(const bf16_t* Q, const bf16_t* X, bf16_t* P, int B, int R, int S, int E, hipStream_t st,
                         const int32_t* cu = nullptr, int H = 0);
int xattn_dscores(const bf16_t* dO, const bf16_t* X, const bf16_t* P, const float* delta, bf16_t* dS, int B, int R, int S, int E,
                  hipStream_t st, const int32_t* cu = nullptr, int H = 0);
// total_rows: sum of the samples' rows (packed: T * H; 0 = dense, B * R) - picks the row tile of the apply kernel
int xattn_apply(const bf16_t* A, const bf16_t* X, bf16_t* out, int B, int R, int S, int E, hipStream_t st, const int32_t* cu = nullptr,
                int H = 0, int64_t total_rows = 0);
// xattn2.hip: the 8-wave second generation of the three large products (SPN_XATTN_V2=0: first generation)
bool xattn2_on();
bool xattn2_apply_ok(int E);
int xattn2_scores_softmax(const bf16_t* Q, const bf16_t* X, bf16_t* P, int B, int R, int S, int E, hipStream_t st, const int32_t* cu, int H);
int xattn2_dscores(const bf16_t* dO, const bf16_t* X, const bf16_t* P, const float* delta, bf16_t* dS, int B, int R, int S, int E,
                   hipStream_t st, const int32_t* cu, int H);
int xattn2_apply(const bf16_t* A, const bf16_t* X, bf16_t* out, int B, int R, int S, int E, hipStream_t st, const int32_t* cu, int H);
int xattn_delta(const bf16_t* dctx, const bf16_t* ctx, const float* bv, float* delta, int T, int H, hipStream_t st);
int xattn_wgrad(const bf16_t* q, size_t q_stride, const bf16_t* dqa, size_t dqa_stride, const bf16_t* dctx, size_t dctx_stride,
                const bf16_t* oa, size_t oa_stride, float* dW, float* dbias, size_t g_stride, int layers, int T, int W, int H, int E,
                float scale, hipStream_t st);

// attention_small.hip: whole-head kernels for Lq == Lk <= 128 (attention_fwd/bwd dispatch to them)
bool attention_small_ok(const AttnArgs& a);
int attention_small_fwd(const AttnArgs& a, hipStream_t st);
int attention_small_bwd(const AttnBwdArgs& a, hipStream_t st);
// few queries (Lq <= 64, no causal mask) over many keys: fused dQ / dK / dV in one pass over K and V (needs delta)
bool attention_cross_ok(const AttnArgs& a);
int attention_cross_bwd(const AttnBwdArgs& a, hipStream_t st);
int attention_cross_fwd(const AttnArgs& a, hipStream_t st);
int attention_cross_fwd_blocks(const AttnArgs& a, hipStream_t st);   // non-causal, any Lq: 64-query blocks in grid.y

// bank.hip
int combine_l2norm_fwd(const float* refer_bank, const int64_t* ref_idx, int64_t n_refer, const float* text,
                       float* q_f32, bf16_t* q_bf16, float* inv_norm, int B, int D, int ldq, hipStream_t st);
int combine_l2norm_bwd(const float* q, const float* inv_norm, const float* dq, float* dtext, int B, int D,
                       hipStream_t st, const float* scale_dev = nullptr);
struct BankArgs {
    const bf16_t* q; int ldq;        // [B, D] L2-normalised queries
    const bf16_t* bank;              // [M_local, D] bf16, or e4m3 bytes [M_local, D] when bank_scale != nullptr
    const int64_t* labels;           // [B] global row ids
    int B, M, D;                     // M = rows in this shard
    int m_begin;                     // global id of row 0 of this shard
    float inv_tau;
    const float* bank_scale = nullptr;   // fp8 bank: per-row dequantisation scale [M_local]
    int group = 0;                       // 32: token-max bank - 32 rows per target, logit = max over them; labels,
                                         // m_begin and the softmax run over TARGETS (M stays the row count)
};
// fp32 [M, D] -> OCP e4m3 [M, Dp] (zero padded) with one fp32 scale per row (row max -> 448)
int bank_quantize_fp8(const float* bank, int M, int D, int Dp, uint8_t* out, float* scale, hipStream_t st);
int bank_dequant_fp8(const uint8_t* data, const float* scale, int M, int D, bf16_t* out, hipStream_t st);
// forward: per-row partial softmax statistics over this shard
//   stats[b] = {max, sum exp(l - max), sum l, label logit (or -inf if the label is not in the shard)}
// zsave (optional, bank_saved_bytes()): receives the logits when bank_saved_path(a) - hand it to bank_grad_q as zsaved
int bank_stats_fwd(const BankArgs& a, float* stats /*[B,4]*/, float* ws, size_t ws_bytes, hipStream_t st,
                   float* zsave = nullptr);
int bank_stats_fold(const float* ws, int n, int B, float* stats, hipStream_t st);
// bank2.hip: barrier-free kernels for batches below 128 queries, backward from saved logits
bool bank_saved_path(const BankArgs& a);
int bank_config(int mode);
int bank_mode();
int bank_saved_ld(int M);
size_t bank_saved_bytes(int B, int M);
size_t bank_saved_bytes_any(int B, int M);   // bank.hip: the same incl. the layout of batches >= 128 (p, G^T, tile maxima)
bool bank_saved_path_large(const BankArgs& a);
size_t bank2_workspace_bytes(int B, int M, int D);
int bank2_stats_fwd(const BankArgs& a, float* stats, float* zsave, float* ws, size_t ws_bytes, hipStream_t st);
int bank2_grad_q(const BankArgs& a, const float* zsaved, const float* row_lse, float label_smoothing, int64_t M_total,
                 float grad_scale, float* dq, float* ws, size_t ws_bytes, hipStream_t st);
// the whole loss step of ONE shard that holds the full bank (m_begin = 0), label smoothing 0, in two launches; SPN_ERR_SHAPE
// when the single-pass kernels do not serve the shape (bank_step_ok) - callers then use the three calls
bool bank_step_ok(const BankArgs& a);
int bank_step(const BankArgs& a, float* save, float grad_scale, float* row_lse, float* row_loss, float* loss_mean, float* dq,
              hipStream_t st);
// bank3.hip: forward statistics for 128..256 queries in 160-row bank tiles (one workgroup per CU, deep bank prefetch)
bool bank_stats160_ok(int B, int D, int ldq);
int bank_stats160_tiles(int M);
int bank_stats160(const bf16_t* q, int ldq, const bf16_t* bank, const int64_t* labels, int B, int M, int D, int m_begin, float inv_tau,
                  float* partial, bf16_t* Pt, float* tmax, hipStream_t st, int* tail_counter = nullptr);
int bank_stats_tail(const float* partial, int ntiles, int B, float* row_lse, float* row_loss, float* loss_mean, int* counter,
                    hipStream_t st);
int bank_gt_scale(const bf16_t* Pt, bf16_t* Gt, const float* tmax, const float* lse, const int64_t* labels, int B, int M, int m_begin,
                  float ls, float inv_m, hipStream_t st);
// finalize on the owner of all shards' stats: row_lse, row_loss and the mean loss
int bank_loss_finalize(const float* stats, int nshards, int B, int64_t M_total, float label_smoothing,
                       float* row_lse, float* row_loss, float* loss_mean, hipStream_t st);
// backward: dq[b] (+)= grad_scale * inv_tau * sum_m (softmax - target) bank[m]
int bank_grad_q(const BankArgs& a, const float* row_lse, float label_smoothing, int64_t M_total, float grad_scale,
                float* dq /*[B,D] fp32*/, float* ws, size_t ws_bytes, hipStream_t st, const float* zsaved = nullptr);
size_t bank_workspace_bytes(int B, int M, int D);
size_t bank_workspace_bytes_fp8(int B, int M, int D);   // + room for the expanded bf16 copy at large batches
// in-batch negatives (clip4cir/models.py:160-167): dt[j] = grad_scale * inv_tau * sum_i (softmax_i[j] - [i == j]) q[i]
int inbatch_grad_t(const bf16_t* q, const bf16_t* t, int ldq, const float* row_lse, int B, int D, float inv_tau,
                   float grad_scale, float* dt, hipStream_t st);

// preprocess.hip (mean3 / std3 are HOST pointers)
int preprocess_image(const uint8_t* src, int H, int W, int hp, int vp, const int32_t* kx, const int32_t* bx, int ksize_x,
                     const int32_t* ky, const int32_t* by, int ksize_y, int crop_left, int crop_top, int dim,
                     const float* mean3, const float* std3, uint8_t* tmp, float* out, uint8_t* out_u8, hipStream_t st);

// tgcir.hip (TG-CIR head, tgcir/models.py:21-49,127-151,198-205)
size_t tg_ws_bytes(int B, int C);
int tg_tokenlearn_fwd(const float* z, const float* w, const float* bias, float* A, float* mod, int B, int L, int C, int S,
                      int G, hipStream_t st);
int tg_tokenlearn_bwd(const float* z, const float* w, const float* A, const float* dmod, bf16_t* dz, float* dw, float* dbias,
                      float* ws, size_t ws_bytes, int B, int L, int C, int S, int G, hipStream_t st);
int tg_fuse_prep(const float* feats, const float* masks, const float* ref, float* mod, bf16_t* X, float* Xf, int B, int C,
                 int S, int G, hipStream_t st);
int tg_img_finish(const float* feats, const float* masks, float* mod, float* pooled, int B, int C, int S, int G,
                  hipStream_t st);
int tg_gate_fwd(const float* hpre, const float* w2, const float* b2, const float* ref, const float* mod, float* r,
                float* pooled, int B, int NT, int C, hipStream_t st);
int tg_gate_bwd(const float* dpooled, const float* ref, const float* mod, const float* r, const float* hpre, const float* w2,
                float* dmod, float* dh, float* dhT, float* dw2, float* db1, float* db2, float* ws, size_t ws_bytes, int B,
                int NT, int C, hipStream_t st);
int tg_mod_bwd(const float* dX, float* dmod, const float* feats, const float* masks, float* dfeats, float* dmasks, float* ws,
               size_t ws_bytes, int B, int C, int S, int G, hipStream_t st);

// topk.hip
int cosine_scores_f64(const float* q, const float* gallery, int Nq, int Ng, int D, double* out, hipStream_t st);
int topk_from_scores(const double* scores, int Nq, int Ng, int K, const int32_t* exclude, int32_t* idx, double* val,
                     hipStream_t st);

}  // namespace spn
