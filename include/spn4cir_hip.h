/* spn4cir_hip.h — C-ABI of libspn4cir_hip.so (MI355X / gfx950).
 *
 * The reference (BUAADreamer/SPN4CIR) is pure Python on stock PyTorch ops and has no FFI of
 * its own; the seam this library sits under is the Python `CIRPlus` protocol
 * (clip4cir/models_negplus.py:16-154, zscir/models_bank.py:18-134) and the torch ops it
 * calls.  Each entry point below names the reference code it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = ok, < 0 = argument/shape/workspace error
 *     (SPN_ERR_*), > 0 = hipError_t of a failed launch.  Nothing throws.
 *   - all data pointers are DEVICE pointers owned by the caller; `bf16` buffers are raw
 *     2-byte bfloat16 (void* in the signatures), `float` is fp32.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Calls only enqueue
 *     work; no function synchronises the device or allocates memory.
 *   - no mutable global state: one caller thread per stream, as under one-process-per-GPU DDP.
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef SPN4CIR_HIP_H
#define SPN4CIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPN_ABI_VERSION 1

#define SPN_ERR_ARG (-1)
#define SPN_ERR_SHAPE (-2)
#define SPN_ERR_WORKSPACE (-3)

#define SPN_ACT_NONE 0
#define SPN_ACT_QUICKGELU 1 /* x*sigmoid(1.702x), clip/model.py:166-168 */
#define SPN_ACT_GELU_ERF 2  /* exact GELU, blip4cir/med.py BertIntermediate */

int spn_abi_version(void);
const char* spn_error_string(int code);

/* ---------------------------------------------------------------- GEMM (torch.nn.Linear, `@`)
 * C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]); bf16 inputs, fp32 MFMA accumulation.
 * K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0.  Replaces F.linear inside
 * nn.MultiheadAttention / mlp.c_fc / mlp.c_proj (clip/model.py:175-181,187). */
int spn_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, int act,
                void* out_bf16, float* out_f32, void* pre_act_out_bf16, int ldc, void* stream);
/* out_f32 = resid + A.B^T + bias (residual add fused: `x = x + ...`, clip/model.py:190-191) */
int spn_gemm_nt_resid(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias,
                      const float* resid, int ldr, float* out_f32, void* out_bf16, int ldc, void* stream);
/* out_bf16 = (A.B^T) * act'(pre_act)  (backward of the fused activation) */
int spn_gemm_nt_dact(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const void* pre_act_bf16,
                     int act, void* out_bf16, int ldc, void* stream);
/* C[N1,N2] (fp32) = alpha * A[Kr,N1]^T . B[Kr,N2] (+ C): weight gradients dW = dY^T X (autograd of
 * F.linear).  colsum_out (optional, [N1]) receives sum_k A[k][:], i.e. the bias gradient when A = dY.
 * N1, N2 % 8 == 0.  ws: spn_gemm_tn_workspace_bytes() bytes of scratch. */
int spn_gemm_tn(const void* A, const void* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
                float alpha, int accumulate, float* colsum_out, void* ws, size_t ws_bytes, void* stream);
size_t spn_gemm_tn_workspace_bytes(int Kr, int N1, int N2);
/* Two such products over the same Kr rows in one launch (dense operands: lda = N1, ldb = N2, ldc = N2; both outputs
 * overwritten): a weight gradient with a small output (the W x W out-projection) shares the grid of a larger one (the
 * 3W x W qkv projection) instead of under-filling the chip in a launch of its own. */
int spn_gemm_tn_pair(const void* A1, const void* B1, int N1a, int N2a, float* C1, float* colsum1, const void* A2,
                     const void* B2, int N1b, int N2b, float* C2, float* colsum2, int Kr, void* ws, size_t ws_bytes,
                     void* stream);
size_t spn_gemm_tn_pair_workspace_bytes(int Kr, int N1a, int N2a, int N1b, int N2b);
/* n (<= 48) such products over the same Kr rows in ONE launch WITHOUT split-K: every 256x256 output tile is computed by one
 * workgroup that loops over all Kr rows and writes the final fp32 values (C_p overwritten; colsum_p = column sums of A_p
 * or NULL); only the tiles of a last, mostly empty round of workgroups are split over the reduction (slabs in ws).  This is
 * how a whole backward pass computes its weight gradients (spn_text_bwd_wgrad): autograd's per-layer dW = dY^T X products
 * (train_negplus.py:121, loss.backward()) are independent of each other and of the data path, so they are deferred and
 * batched - no partial sums through HBM, no per-problem reduction launches. */
typedef struct {
    const void* A;      /* bf16 [Kr, lda >= N1] */
    const void* B;      /* bf16 [Kr, ldb >= N2] */
    float* C;           /* fp32 [N1, ldc >= N2] */
    float* colsum;      /* fp32 [N1] or NULL */
    int N1, N2, lda, ldb, ldc;
} spn_tn_problem;
int spn_gemm_tn_grouped(const spn_tn_problem* problems, int n, int Kr, void* ws, size_t ws_bytes, void* stream);
size_t spn_gemm_tn_grouped_workspace_bytes(int Kr);

/* ---------------------------------------------------------------- elementwise / reductions */
int spn_cast_f32_bf16(const float* x, void* y_bf16, size_t n, void* stream);
/* y = bf16(x) and yt = bf16(x)^T for x [rows, cols]; either output may be NULL */
int spn_cast_bf16_f32(const void* x_bf16, float* y, size_t n, void* stream);
/* bf16 gradient exchange between data-parallel replicas (the reference is single-GPU; SURVEY 8d prices the gradient all-reduce at
 * 247 MB in bf16 against 494 MB in fp32): chunks_bf16 [n_ranks][m] = every rank's bf16 copy of ONE slice of the flat gradient (as an
 * all-to-all delivers them), out_bf16 [m] = bf16(sum over the ranks in rank order, accumulated in fp32) - the same bits on every
 * rank.  m % 8 == 0, 16-byte aligned buffers. */
int spn_sum_ranks_bf16(const void* chunks_bf16, int n_ranks, size_t m, void* out_bf16, void* stream);
/* The fp32 flavour (the "direct reduce-scatter + all-gather that drives all 7 links concurrently" of SURVEY section 5, built from an
 * all-to-all and an all-gather): out[i] = sum over the ranks, in rank order, of chunks[r][i]; m % 4 == 0, 16-byte aligned. */
int spn_sum_ranks_f32(const float* chunks, int n_ranks, size_t m, float* out, void* stream);
int spn_cast_transpose_f32_bf16(const float* x, void* y_bf16, void* yt_bf16, int rows, int cols, void* stream);
/* out[c] (+)= sum_r x[r][c]  (bias gradients) */
int spn_colsum_bf16(const void* x_bf16, int rows, int cols, int ld, float* out, int accumulate, void* ws,
                    size_t ws_bytes, void* stream);
size_t spn_colsum_workspace_bytes(int rows, int cols);

/* ---------------------------------------------------------------- LayerNorm (clip/model.py:157-163)
 * fp32 statistics over fp32 input; outputs bf16 and/or fp32. W % 4 == 0, W <= 2048. */
int spn_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                      float* rstd, int rows, int W, float eps, void* stream);
int spn_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* gamma, const float* mean,
                      const float* rstd, float* dx, int accumulate_dx, void* dx_bf16, float* dgamma, float* dbeta,
                      int accumulate_dparam, int rows, int W, void* ws, size_t ws_bytes, void* stream);
size_t spn_layernorm_bwd_workspace_bytes(int rows, int W);

/* ---------------------------------------------------------------- attention core, head_dim 64
 * softmax(scale * q k^T + causal mask + key_bias) v per (batch, head); q rows b*Lq+i, k/v rows
 * b*Lk+j, head h at column h*64.  Replaces the core of nn.MultiheadAttention
 * (clip/model.py:186-187 with the -inf causal mask of :330-336) and BertSelfAttention
 * (blip4cir/med.py:196-234; key_bias = the (1-mask)*-10000 extended mask). */
int spn_attention_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o, int ldo,
                      float* lse, const float* key_bias, int B, int H, int Lq, int Lk, int causal, float scale,
                      void* stream);
int spn_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const void* o, int ldo,
                      const float* lse, const float* key_bias, const void* d_o, int lddo, void* dq, void* dk, void* dv,
                      int lddq, int lddk, int lddv, float* delta_ws, int B, int H, int Lq, int Lk, int causal,
                      float scale, void* stream);

/* ---------------------------------------------------------------- embedding (clip/model.py:346-348) */
int spn_embed_fwd(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* x, int B, int L, int W,
                  int vocab, void* stream);
int spn_embed_bwd(const int32_t* ids, const int32_t* eot_or_null, const float* dx, float* dtok_zeroed, float* dpos,
                  int B, int L, int W, int vocab, void* stream);

/* ---------------------------------------------------------------- combiner + normalise
 * q = F.normalize(refer_bank[ref_idx] + text)   (models_negplus.py:48-50,133-137).
 * refer_bank may be NULL (q = normalize(text)).  q_bf16 has leading dim ldq >= D (pad is zeroed).
 * n_refer = rows of refer_bank: a ref_idx outside [0, n_refer) is never dereferenced (the reference raises
 * IndexError at models_negplus.py:133; a kernel cannot) - that query row and its inv_norm come out as NaN, so the
 * loss of the step is NaN instead of silently wrong.  n_refer <= 0 disables the check. */
int spn_combine_l2norm_fwd(const float* refer_bank, const int64_t* ref_idx, int64_t n_refer, const float* text,
                           float* q_f32, void* q_bf16, float* inv_norm, int B, int D, int ldq, void* stream);
int spn_combine_l2norm_bwd(const float* q_f32, const float* inv_norm, const float* dq, float* dtext, int B, int D,
                           void* stream);
/* the same with every output multiplied by *scale_dev (1-element device fp32): the d(loss) that autograd hands
 * `loss.backward()` / GradScaler's scaled loss (train_negplus.py:121) enters here without a host synchronisation */
int spn_combine_l2norm_bwd_scaled(const float* q_f32, const float* inv_norm, const float* dq, const float* scale_dev,
                                  float* dtext, int B, int D, void* stream);

/* ---------------------------------------------------------------- bank InfoNCE
 * models_negplus.py:150-154: logits = (q @ bank.T)/tau ; CrossEntropyLoss(logits, labels).
 * The bank (or this rank's shard of it: rows [m_begin, m_begin+M) of the global bank) is a
 * device-resident bf16 [M, D] matrix of L2-normalised rows; D in {128,256,512,640,768,1024}.
 *   stats[b] = {max_j l_bj, sum_j exp(l_bj - max), sum_j l_bj, l_b,label (or -inf)} over the shard. */
int spn_bank_stats_fwd(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B, int M, int D,
                       int m_begin, float inv_tau, float* stats, void* ws, size_t ws_bytes, void* stream);
/* combine nshards stats blocks [nshards][B][4] -> row_lse[B], row_loss[B], loss_mean[1] */
int spn_bank_loss_finalize(const float* stats, int nshards, int B, int64_t M_total, float label_smoothing,
                           float* row_lse, float* row_loss, float* loss_mean, void* stream);
/* dq[b,:] = grad_scale * inv_tau * sum_j (softmax_bj - target_bj) bank[j,:] over the shard
 * (grad_scale = GradScaler scale / B_global for the mean loss) */
int spn_bank_grad_q(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B, int M, int D,
                    int m_begin, float inv_tau, const float* row_lse, float label_smoothing, int64_t M_total,
                    float grad_scale, float* dq, void* ws, size_t ws_bytes, void* stream);
size_t spn_bank_workspace_bytes(int B, int M, int D);
/* The forward/backward PAIR of one step: logits_save is spn_bank_logits_bytes(B, M) bytes of device scratch written by the
 * forward call and read by the backward call of the SAME (q, bank, labels, B, M, D, m_begin, inv_tau) - what autograd keeps
 * for `(q @ bank.T) / tau` (models_negplus.py:150-153), in whichever form the shape's kernels want:
 *   default below 192 queries (any B the GEMM pair does not take; e4m3 banks below 256 queries): ONE pass over the bank.  The
 *     forward call computes the statistics AND the unnormalised query gradient sum_j 2^(z_j log2e - r) bank_j per bank chunk
 *     (flash-attention recurrence; scratch = the chunk partials); the backward call folds them with row_lse and subtracts
 *     the label row - it does not read the bank again.  With label_smoothing != 0 the backward call recomputes instead.
 *   B >= 192 (B % 8 == 0, D >= 512, bf16 bank): the forward GEMM's epilogue keeps p = exp(logit - tile max) in bf16 and the
 *     backward pass is a transpose-and-scale launch (G^T) plus ONE weight-gradient-shaped GEMM dq = (G^T)^T bank.
 *   spn_bank_config(1), B < 128: barrier-free streaming kernels (csrc/bank2.hip), fp32 logits kept (B * M * 4 B).
 *   otherwise (token-max banks, e4m3 banks at B >= 256): the pair runs the two calls above - same results, logits_save
 *     untouched.
 * bank_scale = NULL: bf16 bank; else the e4m3 bank of spn_bank_quantize_fp8 (ws from spn_bank_workspace_bytes_fp8). */
size_t spn_bank_logits_bytes(int B, int M);
/* Routing of the pair, process-wide like the SPN_* environment knobs; every mode gives the same results within the
 * documented tolerances: 0 = default (above); 1 = second-generation streaming kernels below 128 queries (also SPN_BANK2=1;
 * slower at 40 000-row banks); 2 = the fused single pass at every batch size (SPN_BANK_FUSED_LARGE=1);
 * 3 = two passes everywhere (SPN_BANK_FUSED=0); 4 = default routing, but the fused pass over an e4m3 bank on the kernel that
 * dequantises each tile into a bf16 image for the dq GEMM (otherwise only taken beyond 2 048 rows per chunk).
 * Mode 1 exists only in the experiments build (make EXPERIMENTS=1 -> libspn4cir_hip_exp.so); the shipped library returns
 * SPN_ERR_ARG for it. */
int spn_bank_config(int mode);
/* Process-wide kernel selection of the GEMMs, for A/B runs and tests (same results either way): key 0 = the persistent
 * multi-round NT kernel (value 1 on, 0 off, -1 back to the SPN_GEMM_PERSIST default; measured slower in the step, so it exists
 * only in the experiments build - the shipped library answers SPN_ERR_ARG to value 1).  SPN_ERR_ARG for an unknown key. */
int spn_gemm_config(int key, int value);
/* Every SPN_* environment variable is captured once, when the library is loaded; the kernels' A/B switches read that
 * snapshot only.  Writes a JSON object {"experiments_build": 0|1, "env": {"SPN_X": "value", ...}} (NUL-terminated, truncated
 * to cap) and returns the size needed.  No reference counterpart (the reference has no native code). */
int spn_config_dump(char* buf, int cap);
int spn_bank_stats_fwd_save(const void* q_bf16, int ldq, const void* bank, const float* bank_scale, const int64_t* labels, int B,
                            int M, int D, int m_begin, float inv_tau, float* stats, float* logits_save, void* ws,
                            size_t ws_bytes, void* stream);
int spn_bank_grad_q_saved(const void* q_bf16, int ldq, const void* bank, const float* bank_scale, const int64_t* labels, int B,
                          int M, int D, int m_begin, float inv_tau, const float* logits_saved, const float* row_lse,
                          float label_smoothing, int64_t M_total, float grad_scale, float* dq, void* ws, size_t ws_bytes,
                          void* stream);
/* The whole loss step of clip4cir/models_negplus.py:150-154 (forward) and its autograd gradient w.r.t. the queries in ONE
 * call, for a bank held entirely by this process (one shard: labels are rows of `bank`, M = M_total) and label smoothing 0:
 * the single pass over the bank followed by ONE tail launch that folds the chunk statistics and partials - row_lse [B],
 * row_loss [B], loss_mean [1] (summed in index order: bit-reproducible) and dq [B, D] fp32 = grad_scale * d(sum of row
 * losses)/dq.  Replaces spn_bank_stats_fwd_save + spn_bank_loss_finalize + spn_bank_grad_q_saved (four launches, three of
 * them launch-bound) where spn_bank_step_ok(B, M, D, fp8) is 1; otherwise SPN_ERR_SHAPE and the caller uses the three calls.
 * logits_save: spn_bank_logits_bytes(B, M) bytes of device scratch. */
int spn_bank_step_ok(int B, int M, int D, int fp8);
int spn_bank_step(const void* q_bf16, int ldq, const void* bank, const float* bank_scale, const int64_t* labels, int B, int M,
                  int D, float inv_tau, float grad_scale, float* logits_save, float* row_lse, float* row_loss, float* loss_mean,
                  float* dq, void* stream);
/* Negative-type ablation head, clip4cir/models_negtype.py:53-134 (text_neg_loss, refer_neg_loss, query_neg_loss, infonce_loss and
 * forward()'s neg_type mask): refer / text / target = fp32 [B, D] raw reference-image, text and target-image features (D a
 * multiple of 64, <= 1024).  With q_i = n(refer_i + text_i), t_i = n(target_i): target term L_ij = <q_i, t_j> (bit 4), query
 * <t_i, q_j> (bit 8), text <n(refer_i + text_j), t_i> (bit 2), refer <n(refer_j + text_i), t_i> (bit 1), each / tau and CE over
 * j with label i; loss [1] = mean of the selected terms, d_refer / d_text / d_target [B, D] = its gradient.  fp32 throughout,
 * every sum in a fixed order.  ws: spn_negtype_workspace_bytes(B, D). */
size_t spn_negtype_workspace_bytes(int B, int D);
int spn_negtype_head(const float* refer, const float* text, const float* target, int B, int D, float inv_tau, int neg_type,
                     float* loss, float* d_refer, float* d_text, float* d_target, void* ws, size_t ws_bytes, void* stream);
/* fp8 bank (BASELINE config 5): the static bank stored as OCP e4m3 bytes [M, Dp] + one fp32 scale per row
 * (row max -> 448), halving the only HBM stream of the loss.  The kernels dequantise a tile into LDS (x scale,
 * bf16) and run the same bf16 MFMA path with fp32 accumulation; q stays bf16.  Same statistics / finalize /
 * dq contract as the bf16 entry points above. */
int spn_bank_quantize_fp8(const float* bank_f32, int M, int D, int Dp, void* bank_fp8, float* bank_scale, void* stream);
/* bank_bf16 [M, D] = bf16(e4m3 x row scale): the image the fp8 calls expand into scratch once per pass at >= 128 queries.  A
 * caller that scores >= 256 queries per call every step (config 5 at its per-GPU batch of 256: the pass is MFMA-bound there,
 * not byte-bound) keeps this image next to the e4m3 bytes and passes it to the bf16 entry points: same values, no expansion
 * per pass, and the saved-probabilities backward (spn_bank_stats_fwd_save / spn_bank_grad_q_saved) applies.  No reference
 * counterpart (zscir/train_bank.py holds an fp32 bank). */
int spn_bank_dequant_fp8(const void* bank_fp8, const float* bank_scale, int M, int D, void* bank_bf16, void* stream);
/* workspace for the two fp8 calls: at B >= 128 it includes room for a bf16 expansion of the shard, made once per
 * pass (every bank tile then serves many query tiles; the per-tile dequantisation only pays at small batches).  With
 * only spn_bank_workspace_bytes() the calls still work and dequantise per tile. */
size_t spn_bank_workspace_bytes_fp8(int B, int M, int D);
int spn_bank_stats_fwd_fp8(const void* q_bf16, int ldq, const void* bank_fp8, const float* bank_scale,
                           const int64_t* labels, int B, int M, int D, int m_begin, float inv_tau, float* stats, void* ws,
                           size_t ws_bytes, void* stream);
int spn_bank_grad_q_fp8(const void* q_bf16, int ldq, const void* bank_fp8, const float* bank_scale, const int64_t* labels,
                        int B, int M, int D, int m_begin, float inv_tau, const float* row_lse, float label_smoothing,
                        int64_t M_total, float grad_scale, float* dq, void* ws, size_t ws_bytes, void* stream);
/* Token-max bank (blip24cir/lavis/models/blip2_models/blip2_qformer_cir_align_prompt.py:253-265,
 * forward_stage2): every target is 32 Q-Former token rows, bank bf16 [n_targets * 32, D] (target t = rows 32t..32t+31),
 * logit(b, t) = max_k q_b . bank[32t + k] * inv_tau, cross entropy over the targets.  labels / t_begin /
 * targets_total count TARGETS.  The reference loops over the batch in Python with one [M,32] matmul per sample;
 * here the bank streams through LDS once per pass, the max is taken per 32-row tile and the gradient goes to the
 * arg-max row (first index on ties, as torch.max).  Same statistics / finalize contract as above; workspace from
 * spn_bank_workspace_bytes(B, n_targets * 32, D).  One call addresses its bank shard with 32-bit offsets:
 * n_targets * 32 * D * 2 bytes < 4 GiB (262 143 targets at D = 256); larger banks are split into shards (t_begin). */
int spn_bank_stats_fwd_tokmax(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B,
                              int n_targets, int D, int t_begin, float inv_tau, float* stats, void* ws, size_t ws_bytes,
                              void* stream);
int spn_bank_grad_q_tokmax(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B, int n_targets,
                           int D, int t_begin, float inv_tau, const float* row_lse, float label_smoothing,
                           int64_t targets_total, float grad_scale, float* dq, void* ws, size_t ws_bytes, void* stream);
/* In-batch negatives, BASELINE config 1 (clip4cir/models.py:151-167, wo_bank: labels = arange(B), the target
 * features are trainable too).  Loss and dq come from the three calls above with the normalised target
 * features as the bank; this is the target-side gradient
 *   dt[j,:] = grad_scale * inv_tau * sum_i (exp(q_i.t_j * inv_tau - row_lse[i]) - [i == j]) q[i,:]   fp32 [B, D] */
int spn_inbatch_grad_t(const void* q_bf16, const void* t_bf16, int ld, const float* row_lse, int B, int D,
                       float inv_tau, float grad_scale, float* dt, void* stream);

/* ---------------------------------------------------------------- AdamW (train_negplus.py:77-83)
 * torch.optim.AdamW semantics; g is multiplied by inv_scale (GradScaler.unscale_), the step is
 * skipped when *found_inf != 0 (GradScaler.step).  step is 1-based. */
int spn_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float inv_scale, const float* found_inf, void* stream);
/* the same with the gradient (loss) scale read on the DEVICE: g is divided by *grad_scale (NULL = 1) - what
 * torch.amp.GradScaler hands an optimizer that declares _step_supports_amp_scaling (optimizer.grad_scale /
 * optimizer.found_inf, both 1-element device tensors), so the reference's scaler.step(optimizer) needs no host sync. */
int spn_adamw_step_scaled(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, const float* grad_scale, const float* found_inf,
                          void* stream);
/* the same with the STEP COUNT on the device as well: step_dev is a 1-element fp32 device counter advanced by
 * spn_adamw_tick once per optimizer step (before the spn_adamw_step_dev launches of that step) unless *found_inf != 0,
 * so a step GradScaler skips does not advance the bias correction - torch does not call optimizer.step() on overflow
 * (torch/amp/grad_scaler.py: _maybe_opt_step), its fused path subtracts found_inf from the per-parameter step. */
int spn_adamw_tick(float* step_dev, const float* found_inf, void* stream);
int spn_adamw_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, const float* step_dev, const float* grad_scale, const float* found_inf,
                       void* stream);
int spn_grad_check_finite(const float* g, size_t n, float* found_inf, void* stream);

/* ---------------------------------------------------------------- Recall@K (validate.py:28-33)
 * scores[i][j] = <q_i, g_j> accumulated in fp64; top-K by (score desc, index asc), skipping
 * exclude[i] (the reference image, validate.py:39,130-134) when exclude != NULL. */
int spn_cosine_scores_f64(const float* q, const float* gallery, int Nq, int Ng, int D, double* scores, void* stream);
int spn_topk_from_scores(const double* scores, int Nq, int Ng, int K, const int32_t* exclude, int32_t* idx,
                         double* val, void* stream);

/* ---------------------------------------------------------------- CLIP text tower
 * CLIP.encode_text (clip/model.py:345-358) forward and its backward, 12 x
 * ResidualAttentionBlock (:171-192).  Parameters live in ONE flat fp32 buffer laid out by
 * spn_text_layout(); the GEMM weights are mirrored (plus transposes) in a flat bf16 buffer
 * that spn_text_refresh_bf16() rewrites after every optimizer step. */
typedef struct {
    int B, L, L_ctx, W, H, layers, D, vocab;
    /* T = 0: dense, every caption occupies L rows (what clip/model.py:345-358 computes).
     * T > 0: packed, only the T = sum_b (eot_b + 1) live rows exist.  Rows after a caption's EOT token are dead
     * under the causal mask (clip/model.py:330-336): they feed neither x[arange, argmax] (:356) nor any gradient,
     * so dropping them changes no output.  Use spn_text_fwd_packed(); the other calls take the same cfg. */
    int T;
    /* pool != 0: the LAST block runs its row-wise half - out-projection, ln_2, MLP, and their backward with the three weight
     * gradients - on the B pooled rows only.  clip/model.py:352-356 reads one row per caption (x[arange, argmax]) behind
     * ln_final, so no other row of the last block's output reaches the feature or any gradient; the rows are computed by the
     * same kernels, the result is that of the all-rows computation.  Forward and backward calls must agree on it; the token
     * variants (spn_text_fwd_tokens / spn_text_bwd_tokens*), which need ln_final of every row, require 0. */
    int pool;
} spn_text_cfg;

typedef struct {
    int64_t tok, pos, blocks, block_size, lnf_g, lnf_b, text_proj, n_params;
    /* inside a block: ln1_g ln1_b w_qkv b_qkv w_o b_o ln2_g ln2_b w_fc b_fc w_proj b_proj, [12] = size */
    int64_t block_off[13];
    int64_t bf16_block_size, bf16_text_proj, bf16_text_proj_t, n_bf16;
} spn_text_layout_t;

int spn_text_layout(const spn_text_cfg* cfg, spn_text_layout_t* out);
size_t spn_text_act_bytes(const spn_text_cfg* cfg);
size_t spn_text_ws_bytes(const spn_text_cfg* cfg);
int spn_text_refresh_bf16(const spn_text_cfg* cfg, const float* params, void* weights_bf16, void* stream);
int spn_text_fwd(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                 void* acts, float* feats, void* stream);
/* cfg->T > 0; cu_seqlens = device int32 [B+1], cu[0] = 0, cu[b+1] - cu[b] = argmax_l ids[b, l] + 1, cu[B] = T.
 * It is copied into acts, so the backward calls below need nothing extra.  L <= 128. */
int spn_text_fwd_packed(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        const int32_t* cu_seqlens, void* acts, float* feats, void* stream);
int spn_text_bwd(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                 void* acts, const float* dfeats, float* grads, void* ws, size_t ws_bytes, void* stream);
/* The same backward in three phases (head, layers-1..0, tail) so a data-parallel host can start
 * the gradient all-reduce of a layer's span as soon as that layer's call has been enqueued
 * (autograd's reverse-order hooks in DDP).  ws must be the same buffer in every phase. */
int spn_text_bwd_head(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                      const float* dfeats, float* grads, void* ws, size_t ws_bytes, void* stream);
int spn_text_bwd_layer(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                       float* grads, int layer, void* ws, size_t ws_bytes, void* stream);
int spn_text_bwd_tail(const spn_text_cfg* cfg, const int32_t* ids, void* acts, float* grads, void* ws,
                      size_t ws_bytes, void* stream);
/* Deferred weight gradients.  spn_text_bwd_layer_deferred = spn_text_bwd_layer without the block's four weight-gradient
 * products dW = dY^T X (c_proj, c_fc, in_proj, out_proj: nothing downstream reads them): their dY operands stay in
 * per-layer buffers of ws, and spn_text_bwd_wgrad computes them for the blocks [layer_begin, layer_end) (<= 12) in ONE
 * grouped launch (spn_gemm_tn_grouped) after ALL of those blocks have gone through the deferred call.  LayerNorm
 * gradients of a block are final after its own call, weights and biases after the wgrad call: a data-parallel host
 * starts a span's all-reduce behind the latter.  spn_text_bwd defers every block by itself. */
int spn_text_bwd_layer_deferred(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                                float* grads, int layer, void* ws, size_t ws_bytes, void* stream);
int spn_text_bwd_wgrad(const spn_text_cfg* cfg, void* acts, float* grads, int layer_begin, int layer_end, void* ws,
                       size_t ws_bytes, void* stream);

/* Token output for TG-CIR (tgcir/models.py:127-151, Backbone.extract_text_fea): as spn_text_fwd, plus
 * tokens [B*L, W] fp32 (and an optional bf16 copy) = ln_final of EVERY row - TG-CIR feeds all 77 positions, padding
 * included, to text_fc + TokenLearner - with the per-row statistics tok_mean / tok_rstd [B*L] the backward needs.
 * Dense layout only (cfg->T == 0).  spn_text_bwd_tokens takes the gradients of both outputs. */
int spn_text_fwd_tokens(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        void* acts, float* feats, float* tokens, void* tokens_bf16, float* tok_mean, float* tok_rstd,
                        void* stream);
int spn_text_bwd_tokens(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        void* acts, const float* dfeats, const float* dtokens, const float* tok_mean, const float* tok_rstd,
                        float* grads, void* ws, size_t ws_bytes, void* stream);

/* spn_text_bwd_tokens in phases, for a data-parallel TG-CIR step (tgcir/train.py has no DDP): spn_text_bwd_tokens_head
 * (the pooled feature's and every token's gradient into the residual gradient + ln_final's gradients), then
 * spn_text_bwd_layer_deferred / spn_text_bwd_wgrad over layer groups as for spn_text_bwd, then spn_text_bwd_tail_tokens (the
 * embedding gradients of EVERY position - the padding rows are live here).  Same ws in every phase. */
int spn_text_bwd_tokens_head(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                             const float* dfeats, const float* dtokens, const float* tok_mean, const float* tok_rstd,
                             float* grads, void* ws, size_t ws_bytes, void* stream);
int spn_text_bwd_tail_tokens(const spn_text_cfg* cfg, const int32_t* ids, void* acts, float* grads, void* ws, size_t ws_bytes,
                             void* stream);

/* TG-CIR second-stage head (tgcir/models.py): the per-sample glue between spn_text_fwd_tokens and the
 * bank InfoNCE calls.  All buffers fp32 unless named *_bf16; S = 8 local tokens, G global tokens, NT = G + S <= 16,
 * L <= 640.  The two Linear layers with a GEMM shape (Backbone.text_fc, s_remain_map[0]) are spn_gemm_nt / spn_gemm_tn
 * calls made by the host between these (s_remain_map[0] on spn_gemm_f32: its gradient sums cancel heavily across
 * tokens and do not survive bf16 operands).  ws: spn_tg_ws_bytes(B, C).
 *   tokenlearn (models.py:21-49 SpatialAttention / TokenLearner, applied to z = text_fc(tokens) at :149):
 *       attn[b,l,s] = sigmoid(<z[b,l,:], w[s,:]> + bias[s]);  mod_tokens[b, G+s, :] = mean_l attn[b,l,s] z[b,l,:]
 *   fuse_prep (models.py:136-148 + the cat of :200): mod_tokens[b, i<G, :] = feats[b,:] * relu(masks[i,:]);
 *       x_bf16 / x_f32[b*NT+t, :] = cat(ref_tokens[b,t,:], mod_tokens[b,t,:])
 *   gate (models.py:198-204, img_txt_fusion): remain = sigmoid(<relu(hpre), w2> + b2), hpre = s_remain_map[0](x);
 *       pooled[b,:] = mean_t remain * ref + (1 - remain) * mod     (the caller L2-normalises: spn_combine_l2norm_fwd)
 *   the *_bwd calls return the gradients of every input named d*. */
size_t spn_tg_ws_bytes(int B, int C);
int spn_tg_tokenlearn_fwd(const float* z, const float* w, const float* bias, float* attn, float* mod_tokens, int B, int L,
                          int C, int S, int G, void* stream);
int spn_tg_tokenlearn_bwd(const float* z, const float* w, const float* attn, const float* dmod_tokens, void* dz_bf16,
                          float* dw, float* dbias, void* ws, size_t ws_bytes, int B, int L, int C, int S, int G,
                          void* stream);
int spn_tg_fuse_prep(const float* feats, const float* masks, const float* ref_tokens, float* mod_tokens, void* x_bf16,
                     float* x_f32 /* either may be NULL */, int B, int C, int S, int G, void* stream);
/* image side (models.py:84-125,183-196): tokens[b, i<G, :] = feats[b,:] * relu(masks[i,:]) next to the local tokens
 * spn_tg_tokenlearn_fwd wrote at G.., pooled[b,:] = mean_t tokens[b,t,:] (the caller L2-normalises) */
int spn_tg_img_finish(const float* feats, const float* masks, float* tokens, float* pooled, int B, int C, int S, int G,
                      void* stream);
int spn_tg_gate_fwd(const float* hpre, const float* w2, const float* b2, const float* ref_tokens, const float* mod_tokens,
                    float* remain, float* pooled, int B, int NT, int C, void* stream);
int spn_tg_gate_bwd(const float* dpooled, const float* ref_tokens, const float* mod_tokens, const float* remain,
                    const float* hpre, const float* w2, float* dmod_tokens, float* dh /* [B*NT, C] */,
                    float* dh_t /* [C, B*NT] */, float* dw2, float* db1, float* db2, void* ws, size_t ws_bytes, int B, int NT,
                    int C, void* stream);
int spn_tg_mod_bwd(const float* dx, float* dmod_tokens, const float* feats, const float* masks, float* dfeats, float* dmasks,
                   void* ws, size_t ws_bytes, int B, int C, int S, int G, void* stream);

/* ---------------------------------------------------------------- image preprocessing
 * targetpad_transform (clip4cir/data_utils.py:42-65,84-98): TargetPad -> Resize(dim, BICUBIC) -> CenterCrop(dim) ->
 * ToTensor -> Normalize on a decoded RGB image, bit-identical to the Pillow/torchvision pipeline the reference
 * runs on the CPU.  src_rgb: device uint8 [H, W, 3]; pad_x / pad_y: TargetPad's zero borders (virtual);
 * kx [ow, ksize_x] int32 22-bit fixed-point coefficients and bx [ow, 2] = (first source column, count) of the
 * horizontal pass over the PADDED width, ky / by likewise for the vertical pass (host-computed, Pillow's
 * precompute_coeffs); crop_left / crop_top: CenterCrop offsets in the resized image; tmp: device scratch
 * (H + 2 pad_y) * dim * 3 bytes; out_chw: fp32 [3, dim, dim] (may be NULL), out_u8_hwc: the uint8 crop before
 * ToTensor [dim, dim, 3] (may be NULL); mean3 / std3: HOST float[3]. */
int spn_preprocess_image(const uint8_t* src_rgb, int H, int W, int pad_x, int pad_y, const int32_t* kx, const int32_t* bx,
                         int ksize_x, const int32_t* ky, const int32_t* by, int ksize_y, int crop_left, int crop_top, int dim,
                         const float* mean3, const float* std3, void* tmp, float* out_chw, uint8_t* out_u8_hwc,
                         void* stream);

/* ---------------------------------------------------------------- JPEG decode (baseline, batched)
 * Replaces the host decode `PIL.Image.open(path)` + `.convert("RGB")` of the reference's datasets
 * (clip4cir/data_utils_negplus.py:17,268-319) for the calls that decode whole galleries - the bank builders
 * (models_negplus.py:59-125) and extract_index_features (utils.py:24-50): file bytes in, RGB uint8 [H][W][3] in device memory
 * out (what spn_preprocess_image consumes), bit for bit what Pillow / libjpeg-turbo produce at their default settings (islow IDCT,
 * fancy upsampling).  Scope: baseline / extended-sequential Huffman, 8 bit, one interleaved scan, grayscale or YCbCr with chroma
 * 1x1 and luma 1x1 / 2x1 / 2x2; the host parses markers and tables (spn4cir_amd/jpeg.py) and sends anything else to Pillow.
 * spn_jpeg_image: one record per image (offsets into the batch buffers); spn_jpeg_segment: one independently decodable entropy
 * segment (a scan, or one restart interval: byte_off points behind its RSTn marker); spn_jpeg_huff: a Huffman table in decode form
 * (9-bit look-ahead: length << 8 | symbol, 0 = longer code; canonical maxcode / value offsets for lengths 10..16).
 * coefs: int16 scratch of coef_elems elements (zeroed inside the call), planes: uint8 scratch, rgb: output; max_blocks / max_pixels:
 * the largest per-component block count / per-image pixel count of the batch (grid sizing). */
typedef struct {
    int32_t width, height, ncomp, hs, vs, mcux, mcuy, restart_interval;
    uint32_t scan_off, scan_len;
    uint32_t coef_off[3];
    int32_t blocks_x[3], blocks_y[3];
    uint32_t plane_off[3];
    int32_t qt[3], dc_tab[3], ac_tab[3];
    uint32_t rgb_off;
    int32_t first_seg, n_seg;
    int32_t reserved[6];
} spn_jpeg_image;
typedef struct { int32_t image, mcu_first, mcu_count; uint32_t byte_off; } spn_jpeg_segment;
typedef struct { uint16_t look[512]; int32_t maxcode[18]; int32_t valoff[17]; uint8_t sym[256]; uint8_t pad[4]; } spn_jpeg_huff;
int spn_jpeg_decode_batch(const uint8_t* bytes, const spn_jpeg_image* images, int n_images, const spn_jpeg_segment* segments,
                          int n_segments, const spn_jpeg_huff* huff, const uint16_t* qtabs, int16_t* coefs, size_t coef_elems,
                          uint8_t* planes, uint8_t* rgb, int max_blocks, int max_pixels, void* stream);

/* ---------------------------------------------------------------- CLIP vision tower (inference)
 * VisionTransformer.forward (clip/model.py:223-242): frozen in stage 2, used by the bank builders
 * (models_negplus.py:59-125) and extract_index_features (utils.py:24-50).  image: fp32 [B,3,res,res]
 * (already preprocessed); feats: fp32 [B, D] un-normalised.  Flat fp32 parameters per
 * spn_vision_layout(); bf16 mirrors written by spn_vision_refresh_bf16() (once: the tower is frozen). */
typedef struct {
    int B, res, patch, W, H, layers, D;
    int kind; /* 0 = CLIP VisionTransformer; 1 = BLIP / timm ViT (blip4cir/vit.py:115-197: conv bias, no ln_pre,
                 LayerNorm eps 1e-6, exact GELU, final norm over all tokens, vision_proj with bias) */
} spn_vision_cfg;

typedef struct {
    int64_t conv1, conv_b, cls, pos, ln_pre_g, ln_pre_b, blocks, block_size, ln_post_g, ln_post_b, proj, proj_b, n_params;
    int64_t block_off[13];
    int64_t bf16_conv1, bf16_blocks, bf16_block_size, bf16_proj_t, n_bf16, kp, seq;
    int64_t bf16_proj;
} spn_vision_layout_t;

int spn_vision_layout(const spn_vision_cfg* cfg, spn_vision_layout_t* out);
size_t spn_vision_ws_bytes(const spn_vision_cfg* cfg);
int spn_vision_refresh_bf16(const spn_vision_cfg* cfg, const float* params, void* weights_bf16, void* stream);
/* tokens_out (optional, kind 1): the normalised token sequence fp32 [B, S, W] (BLIP_Retrieval.img_embed,
 * blip_cir.py:54-70: the per-image reference bank of blip4cir/models.py:76); feats = proj of token 0. */
int spn_vision_fwd(const spn_vision_cfg* cfg, const float* params, const void* weights_bf16, const float* image,
                   void* ws, size_t ws_bytes, float* feats, float* tokens_out, void* stream);
/* Training path of the CLIP tower (kind 0), used when the visual tower is trainable (clip4cir/models.py:31-33,
 * 156-158: wo_bank / first stage).  The reference re-computes it under torch.utils.checkpoint; here the per-layer
 * activations stay in `acts` (spn_vision_train_act_bytes).  grads: flat fp32, spn_vision_layout() offsets, overwritten. */
size_t spn_vision_train_act_bytes(const spn_vision_cfg* cfg);
size_t spn_vision_bwd_ws_bytes(const spn_vision_cfg* cfg);
int spn_vision_fwd_train(const spn_vision_cfg* cfg, const float* params, const void* weights_bf16, const float* image,
                         void* acts, float* feats, void* stream);
int spn_vision_bwd(const spn_vision_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                   const float* dfeats, float* grads, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------- fp32-exact encode mode (validation)
 * The forward of the two CLIP towers (clip/model.py:223-242, 345-358) with every GEMM on the f32-input MFMA
 * (an f32 fmaf chain), fp32 activations and fp32 attention: for Recall@K validation, where the bf16 towers'
 * feature error (1 - cos ~ 4e-5) is enough to flip near-ties of a ranking.  Reads the fp32
 * parameters directly (no bf16 mirror); inference only; text: dense layout (cfg->T == 0); vision: kind 0. */
size_t spn_text_exact_ws_bytes(const spn_text_cfg* cfg);
int spn_text_fwd_exact(const spn_text_cfg* cfg, const float* params, const int32_t* ids, void* ws, size_t ws_bytes,
                       float* feats, void* stream);
size_t spn_vision_exact_ws_bytes(const spn_vision_cfg* cfg);
int spn_vision_fwd_exact(const spn_vision_cfg* cfg, const float* params, const float* image, void* ws, size_t ws_bytes,
                         float* feats, void* stream);
/* C[M,N] = act(alpha * A[M,K] . op(B) + bias) (+ resid), all fp32; op(B) = B[N,K]^T (b_is_kn = 0) or B[K,N] (1) */
int spn_gemm_f32(const float* A, const float* B, int M, int N, int K, int lda, int ldb, int b_is_kn, const float* bias,
                 int act, const float* resid, int ldr, float* C, int ldc, float alpha, void* stream);

/* ---------------------------------------------------------------- ModifiedResNet image towers (fp32 path)
 * CLIP's RN50 / RN101 / RN50x4 visual towers (clip/model.py:10-155; RN50x4 is train_negplus.py's default model)
 * for encode_image (bank builders, validation).  NHWC fp32 activations; a convolution is spn_im2col3x3_f32 (3x3,
 * padding 1, stride s; 1x1 convolutions need none) + spn_gemm_f32 with the eval-mode BatchNorm folded into weight and
 * bias by the host and act = 3 (ReLU after bias + residual) where the block has one; nn.AvgPool2d(k);
 * AttentionPool2d = spn_attnpool_tokens_f32 (mean token + positional embedding), q/k/v/c projections through
 * spn_gemm_f32 and spn_attnpool_attend_f32 (the pooled token attends to all S = HW + 1 tokens, head_dim 64). */
int spn_im2col3x3_f32(const float* x, float* out, int B, int H, int W, int C, int stride, int nchw, int ldk, void* stream);
int spn_avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int k, void* stream);
int spn_attnpool_tokens_f32(const float* x_nhwc, const float* pos, float* tok, int B, int HW, int C, void* stream);
int spn_attnpool_attend_f32(const float* q, const float* k, const float* v, float* out, int B, int S, int H, void* stream);

/* ModifiedResNet, bf16 fast path (same reference lines; throughput mode for bank extraction / validation, the fp32 calls
 * above stay the exact mode).  NHWC bf16 activations with the channel count padded to a multiple of 64 (Cp; pad channels
 * hold zeros); a 3x3 convolution is spn_im2col3x3_nhwc_bf16 -> [B*Ho*Wo, 9*Cp] (column = tap * Cp + channel) or, for the
 * stem's first convolution on the fp32 NCHW image, spn_im2col3x3_stem_bf16 -> [B*Ho*Wo, 64] (column = tap * 3 + channel,
 * 27 used), followed by spn_gemm_nt (BatchNorm folded into the bf16 weight and the fp32 bias by the host);
 * spn_relu_add_bf16: y = relu(y + resid) in place (resid may be NULL, n % 8 == 0); spn_avgpool_nhwc_bf16 = nn.AvgPool2d(k). */
int spn_im2col3x3_nhwc_bf16(const void* x, void* out, int B, int H, int W, int Cp, int stride, void* stream);
int spn_im2col3x3_stem_bf16(const float* image_nchw, void* out, int B, int H, int W, int stride, void* stream);
int spn_relu_add_bf16(void* y, const void* resid, size_t n, void* stream);
int spn_avgpool_nhwc_bf16(const void* x, void* y, int B, int H, int W, int Cp, int k, void* stream);

/* ---------------------------------------------------------------- BLIP fusion encoder
 * blip4cir/med.py BertModel(mode='multimodal') + text_proj (blip_cir.py:82-98): the query producer of
 * blip4cir/models.py:95-105.  ids [B,L] int32 (ids[:,0] = [ENC]), mask [B,L] int32 {0,1} (may be NULL),
 * enc [B,S,E] fp32 reference-image tokens (constants: no gradient).  proj_out [B,Dp] fp32 = text_proj of
 * the [ENC] position; L2-normalise it with spn_combine_l2norm_fwd(NULL, NULL, proj_out, ...), and feed
 * spn_fusion_bwd the gradient w.r.t. proj_out (spn_combine_l2norm_bwd).  W, E, I, Dp multiples of 64.
 * Per-layer parameter order (layer_off): sa_wqkv[3W,W] (query,key,value rows) sa_bqkv sa_wo sa_bo sa_ln_g
 * sa_ln_b ca_wq ca_bq ca_wkv[2W,E] (key,value rows) ca_bkv ca_wo ca_bo ca_ln_g ca_ln_b ff_w1[I,W] ff_b1
 * ff_w2[W,I] ff_b2 ff_ln_g ff_ln_b, [20] = layer size. */
typedef struct {
    int B, L, S, W, H, layers, I, E, Dp, vocab, max_pos;
    /* Packed text rows (0 = dense).  T > 0: only the T = sum of the caption lengths unmasked positions are materialised - a padded
     * position influences neither the [ENC] feature nor any gradient, because its key is masked in every self-attention
     * (med.py:686 extended_attention_mask) - and the `mask` argument of spn_fusion_fwd / spn_fusion_fwd_bank carries the prefix sums
     * cu_seqlens int32 [B + 1] (cu[0] = 0, cu[B] = T; each length in 1..L; the tokenizer's right padding, blip.py:189-194) instead
     * of the [B, L] mask.  ids stays the padded [B, L] array.  Needs L <= 128 and spn_fusion_packed_ok(cfg) != 0; activation and
     * workspace sizes for T = 0 (the dense rows) reserve the packed form's index arrays and scratch as well, so a buffer sized with
     * T = 0 fits every batch of the shape, packed or dense, up to and including T = B * L (equal-length captions, B = 1). */
    int T;
} spn_fusion_cfg;

typedef struct {
    int64_t word, pos, emb_ln_g, emb_ln_b, layers, layer_size, proj_w, proj_b, n_params;
    int64_t layer_off[21];
    int64_t bf16_layer_size, bf16_proj, bf16_proj_t, n_bf16;
    int64_t bf16_off[15];
} spn_fusion_layout_t;

int spn_fusion_layout(const spn_fusion_cfg* cfg, spn_fusion_layout_t* out);
/* 1 when the shape supports packed text rows (cfg->T is ignored): head width 64, even head count, enc width a multiple of 128,
 * S <= 640, L <= 128 - the absorbed cross-attention of csrc/xattn.hip, which addresses each sample's rows separately. */
int spn_fusion_packed_ok(const spn_fusion_cfg* cfg);
size_t spn_fusion_act_bytes(const spn_fusion_cfg* cfg);
size_t spn_fusion_ws_bytes(const spn_fusion_cfg* cfg);
int spn_fusion_refresh_bf16(const spn_fusion_cfg* cfg, const float* params, void* weights_bf16, void* stream);
int spn_fusion_fwd(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                   const int32_t* mask, const float* enc, void* acts, float* proj_out, void* stream);
/* spn_fusion_fwd with the reference image tokens taken from a DEVICE-RESIDENT bf16 token bank [bank_rows][S][enc_width]
 * (row = one image's token sequence) through token_idx int64 [B]: replaces `self.refer_bank[refer_indexs].detach().to(self.device)`
 * (blip4cir/models.py:97-100: a host gather + a 227 MB upload per B = 128 step; the bank is [N, 577, 768] fp32 in host RAM there,
 * 26.6 GB as bf16 on the device here).  The rows land directly in the bf16 A operand of the cross-attention K/V projections
 * (med.py:178-181).  An index outside [0, bank_rows) yields zero tokens (never dereferenced). */
int spn_fusion_fwd_bank(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        const int32_t* mask, const void* token_bank_bf16, int64_t bank_rows, const int64_t* token_idx, void* acts,
                        float* proj_out, void* stream);
/* The gather alone: out[b, :] = bank[idx[b], :], rows of row_elems bf16 (multiple of 8, 16-byte aligned buffers). */
int spn_gather_bank_rows_bf16(const void* bank_bf16, int64_t bank_rows, const int64_t* idx, void* out_bf16, int B, int64_t row_elems,
                              void* stream);
/* Learnable temperature (blip4cir/models.py:29 `self.tau = nn.Parameter`, :118 `logits = ... / tau`): with dqk = d loss / d (q / tau)
 * [B][lddq] and q fp32 [B][D], writes dtau[0] = -alpha * s * (sum_b <q_b, dqk_b>) / tau^2 (s = scale_dev[0], 1 when NULL: autograd's
 * incoming d(loss); a caller holding d loss / d q instead of d loss / d (q / tau) passes alpha = tau) and inv_tau[0] = 1 / tau
 * (either output may be NULL); tau is read on the device - no host synchronisation; one workgroup, fixed summation order. */
int spn_tau_grad(const float* q, const float* dqk, int lddq, const float* tau_dev, int B, int D, float alpha, const float* scale_dev,
                 float* dtau, float* inv_tau, void* stream);
int spn_fusion_bwd(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                   void* acts, const float* dproj, float* grads, void* ws, size_t ws_bytes, void* stream);

/* The same backward pass in phases, for data-parallel callers that start a layer group's gradient all-reduce while the layers
 * below are still running (blip4cir/train.py has no DDP): phase 0 = head (text_proj; needs dproj), 1 = layers
 * [l_lo, l_hi) from the top down INCLUDING their weight gradients (the flat range of those layers is final when the call
 * returns), 2 = tail (embeddings).  Call 0, then 1 over disjoint groups from layers-1 down to 0, then 2; the residual gradient
 * stays in ws between the calls. */
int spn_fusion_bwd_phase(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                         void* acts, const float* dproj, float* grads, void* ws, size_t ws_bytes, int phase, int l_lo, int l_hi,
                         void* stream);
/* ---------------------------------------------------------------- cross-attention over FROZEN tokens, absorbed form
 * BertSelfAttention with is_cross_attention (blip4cir/med.py:97-181, 196-234): K = X Wk^T + bk and V = X Wv^T + bv are projections of
 * the SAME image tokens X_b [S, E] for all H heads (head width 64), and X carries no gradient (blip4cir/models.py:97-100).  Hence
 *     scores_h = (scale q_h Wk_h) X^T        (bk shifts every score of a query equally: softmax-invariant)
 *     ctx_h    = (softmax(scores_h) X) Wv_h^T + bv_h        (rows of the softmax sum to 1)
 * and K, V [B*S, 2W] are never formed.  q / ctx / dq / dctx: bf16 [rows, W = H*64]; wkv: bf16 [2W, E] (K rows, then V rows), wkv_t its
 * transpose [E, 2W]; bkv fp32 [2W]; x bf16 [B, S, E].  Rows: dense B*L (cu == NULL; row b*L + l) or packed (cu int32 [B+1] prefix
 * sums on the device, T = cu[B] rows, sample b = rows cu[b]..cu[b+1]-1, each length in 1..L).  Caller-provided buffers, kept from
 * forward to backward: qa, oa bf16 [rows, H, E]; p bf16 [rows*H, spn_xattn_sp(S)].  Backward scratch: doa, dqa like qa; ds like p;
 * delta fp32 [rows*H].  Backward writes dq, dwkv fp32 [2W, E] and dbkv fp32 [2W] (the K half of dbkv is exactly zero).
 * spn_xattn_ok: head count even, E % 128 == 0, 1 <= S <= 640 (else SPN_ERR_SHAPE: use K/V projections + spn_attention_*). */
int spn_xattn_ok(int B, int L, int H, int S, int E);
int spn_xattn_sp(int S);
int spn_xattn_fwd(const void* q, const void* wkv, const void* wkv_t, const float* bkv, const void* x, const int32_t* cu, void* qa,
                  void* p, void* oa, void* ctx, int B, int L, int H, int S, int E, int T, float scale, void* stream);
int spn_xattn_bwd(const void* dctx, const void* ctx, const void* q, const void* wkv, const void* wkv_t, const float* bkv,
                  const void* x, const int32_t* cu, const void* p, const void* oa, void* doa, void* ds, void* dqa, float* delta,
                  void* dq, float* dwkv, float* dbkv, int B, int L, int H, int S, int E, int T, float scale, void* stream);

/* out[b, :D] = bf16(x[b, :] * s), out[b, D:ldo] = 0, s = *scale_dev or 1 / *scale_dev (reciprocal != 0): the query scaled by
 * a DEVICE-resident temperature (blip4cir/models.py:29 keeps tau as an nn.Parameter; logits = (q / tau) . bank with the bank
 * calls at inv_tau = 1 - no host read of tau inside a step). */
int spn_scale_cast_bf16(const float* x, const float* scale_dev, int reciprocal, void* out_bf16, int B, int D, int ldo,
                        void* stream);

/* ---------------------------------------------------------------- opt-in kernel timing
 * HIP events recorded on the launch stream around the main kernels (bench.py's live roofline).
 * The only process-global state in the library; off unless spn_prof_enable() is called.
 * kernel ids: 0 gemm_nt, 1 gemm_tn, 2 attention fwd, 3 attention bwd, 4 bank fwd, 5 bank bwd.
 * total_work is FLOPs for ids 0-3 and algorithmic HBM bytes for ids 4-5. */
int spn_prof_enable(int max_records);
int spn_prof_select(unsigned mask, int sample_every); /* bit k = kernel class k; record every n-th launch */
int spn_prof_disable(void);
int spn_prof_reset(void);
int spn_prof_collect(int kernel_id, double* total_ms, double* total_work, int* count);

#ifdef __cplusplus
}
#endif
#endif /* SPN4CIR_HIP_H */
