// extern "C" surface of libspn4cir_hip.so: thin argument marshalling over the C++ kernels.
#include "../../include/spn4cir_hip.h"
#include "tower.h"

using namespace spn;

static_assert(sizeof(spn_text_cfg) == sizeof(TextCfg), "spn_text_cfg layout");
static_assert(sizeof(spn_text_layout_t) == sizeof(TextLayout), "spn_text_layout_t layout");

#define ST(s) ((hipStream_t)(s))
#define BF(p) ((bf16_t*)(p))
#define CBF(p) ((const bf16_t*)(p))

extern "C" {

int spn_abi_version(void) { return SPN_ABI_VERSION; }

const char* spn_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case SPN_ERR_ARG: return "invalid argument";
        case SPN_ERR_SHAPE: return "unsupported shape / alignment";
        case SPN_ERR_WORKSPACE: return "workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int spn_gemm_nt(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias, int act,
                void* out_bf16, float* out_f32, void* pre_act_out_bf16, int ldc, void* stream) {
    GemmEpilogue e;
    e.bias = bias; e.act = act; e.out_bf16 = BF(out_bf16); e.out_f32 = out_f32; e.aux_out = BF(pre_act_out_bf16);
    e.ldc = ldc;
    return gemm_nt(CBF(A), CBF(B), M, N, K, lda, ldb, GEMM_STORE, e, ST(stream));
}

int spn_gemm_nt_resid(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const float* bias,
                      const float* resid, int ldr, float* out_f32, void* out_bf16, int ldc, void* stream) {
    GemmEpilogue e;
    e.bias = bias; e.resid = resid; e.ldr = ldr; e.out_f32 = out_f32; e.out_bf16 = BF(out_bf16); e.ldc = ldc;
    return gemm_nt(CBF(A), CBF(B), M, N, K, lda, ldb, GEMM_RESID, e, ST(stream));
}

int spn_gemm_nt_dact(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const void* pre_act_bf16,
                     int act, void* out_bf16, int ldc, void* stream) {
    GemmEpilogue e;
    e.aux_in = CBF(pre_act_bf16); e.act = act; e.out_bf16 = BF(out_bf16); e.ldc = ldc;
    return gemm_nt(CBF(A), CBF(B), M, N, K, lda, ldb, GEMM_DACT, e, ST(stream));
}

int spn_gemm_tn(const void* A, const void* B, int Kr, int N1, int N2, int lda, int ldb, float* C, int ldc,
                float alpha, int accumulate, float* colsum_out, void* ws, size_t ws_bytes, void* stream) {
    return gemm_tn(CBF(A), CBF(B), Kr, N1, N2, lda, ldb, C, ldc, alpha, accumulate, colsum_out, (float*)ws, ws_bytes,
                   ST(stream));
}

size_t spn_gemm_tn_workspace_bytes(int Kr, int N1, int N2) { return gemm_tn_workspace_bytes(Kr, N1, N2); }

int spn_gemm_tn_pair(const void* A1, const void* B1, int N1a, int N2a, float* C1, float* colsum1, const void* A2,
                     const void* B2, int N1b, int N2b, float* C2, float* colsum2, int Kr, void* ws, size_t ws_bytes,
                     void* stream) {
    if (!A1 || !B1 || !C1 || !A2 || !B2 || !C2 || !ws) return SPN_ERR_ARG;
    return gemm_tn2_pair(CBF(A1), CBF(B1), N1a, N2a, N1a, N2a, C1, N2a, colsum1, CBF(A2), CBF(B2), N1b, N2b, N1b, N2b, C2, N2b,
                         colsum2, Kr, (float*)ws, ws_bytes, ST(stream));
}

size_t spn_gemm_tn_pair_workspace_bytes(int Kr, int N1a, int N2a, int N1b, int N2b) {
    return gemm_tn2_pair_workspace_bytes(Kr, N1a, N2a, N1b, N2b);
}

int spn_gemm_tn_grouped(const spn_tn_problem* problems, int n, int Kr, void* ws, size_t ws_bytes, void* stream) {
    if (!problems || n <= 0 || n > TN_GROUP_MAX) return SPN_ERR_ARG;
    TnProblem p[TN_GROUP_MAX];
    for (int i = 0; i < n; ++i) {
        p[i].A = CBF(problems[i].A); p[i].B = CBF(problems[i].B); p[i].C = problems[i].C; p[i].colsum = problems[i].colsum;
        p[i].N1 = problems[i].N1; p[i].N2 = problems[i].N2; p[i].lda = problems[i].lda; p[i].ldb = problems[i].ldb;
        p[i].ldc = problems[i].ldc;
    }
    return gemm_tn_grouped(p, n, Kr, (float*)ws, ws_bytes, ST(stream));
}

size_t spn_gemm_tn_grouped_workspace_bytes(int Kr) { return gemm_tn_grouped_workspace_bytes(Kr); }

int spn_cast_f32_bf16(const float* x, void* y, size_t n, void* stream) { return cast_f32_bf16(x, BF(y), n, ST(stream)); }

int spn_cast_transpose_f32_bf16(const float* x, void* y, void* yt, int rows, int cols, void* stream) {
    return cast_transpose_f32_bf16(x, BF(y), BF(yt), rows, cols, ST(stream));
}

int spn_colsum_bf16(const void* x, int rows, int cols, int ld, float* out, int accumulate, void* ws, size_t ws_bytes,
                    void* stream) {
    return colsum_bf16(CBF(x), rows, cols, ld, out, accumulate, (float*)ws, ws_bytes, ST(stream));
}

size_t spn_colsum_workspace_bytes(int rows, int cols) { return colsum_workspace_bytes(rows, cols); }

int spn_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32, float* mean,
                      float* rstd, int rows, int W, float eps, void* stream) {
    return layernorm_fwd(x, gamma, beta, BF(y_bf16), y_f32, mean, rstd, rows, W, eps, ST(stream));
}

int spn_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* gamma, const float* mean,
                      const float* rstd, float* dx, int accumulate_dx, void* dx_bf16, float* dgamma, float* dbeta,
                      int accumulate_dparam, int rows, int W, void* ws, size_t ws_bytes, void* stream) {
    return layernorm_bwd(CBF(dy_bf16), dy_f32, x, gamma, mean, rstd, dx, accumulate_dx, BF(dx_bf16), dgamma, dbeta,
                         accumulate_dparam, rows, W, (float*)ws, ws_bytes, ST(stream));
}

size_t spn_layernorm_bwd_workspace_bytes(int rows, int W) { return layernorm_bwd_workspace_bytes(rows, W); }

static AttnArgs make_attn(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o, int ldo,
                          float* lse, const float* key_bias, int B, int H, int Lq, int Lk, int causal, float scale) {
    AttnArgs a;
    a.q = CBF(q); a.k = CBF(k); a.v = CBF(v); a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
    a.o = BF(o); a.ldo = ldo; a.lse = lse; a.key_bias = key_bias;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.causal = causal; a.scale = scale;
    return a;
}

int spn_attention_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* o, int ldo,
                      float* lse, const float* key_bias, int B, int H, int Lq, int Lk, int causal, float scale,
                      void* stream) {
    if (!q || !k || !v || !o) return SPN_ERR_ARG;
    return attention_fwd(make_attn(q, k, v, ldq, ldk, ldv, o, ldo, lse, key_bias, B, H, Lq, Lk, causal, scale),
                         ST(stream));
}

int spn_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const void* o, int ldo,
                      const float* lse, const float* key_bias, const void* d_o, int lddo, void* dq, void* dk, void* dv,
                      int lddq, int lddk, int lddv, float* delta_ws, int B, int H, int Lq, int Lk, int causal,
                      float scale, void* stream) {
    if (!q || !k || !v || !o || !d_o || !dq || !dk || !dv) return SPN_ERR_ARG;
    AttnBwdArgs g;
    g.f = make_attn(q, k, v, ldq, ldk, ldv, const_cast<void*>(o), ldo, const_cast<float*>(lse), key_bias, B, H, Lq, Lk,
                    causal, scale);
    g.d_o = CBF(d_o); g.lddo = lddo;
    g.dq = BF(dq); g.dk = BF(dk); g.dv = BF(dv); g.lddq = lddq; g.lddk = lddk; g.lddv = lddv;
    g.delta = delta_ws;
    return attention_bwd(g, ST(stream));
}

int spn_embed_fwd(const int32_t* ids, const float* tok_emb, const float* pos_emb, float* x, int B, int L, int W,
                  int vocab, void* stream) {
    return embed_fwd(ids, tok_emb, pos_emb, x, B, L, W, vocab, ST(stream));
}

int spn_embed_bwd(const int32_t* ids, const int32_t* eot, const float* dx, float* dtok, float* dpos, int B, int L, int W,
                  int vocab, void* stream) {
    return embed_bwd(ids, eot, dx, dtok, dpos, B, L, W, vocab, ST(stream));
}

int spn_combine_l2norm_fwd(const float* refer_bank, const int64_t* ref_idx, int64_t n_refer, const float* text,
                           float* q_f32, void* q_bf16, float* inv_norm, int B, int D, int ldq, void* stream) {
    return combine_l2norm_fwd(refer_bank, ref_idx, n_refer, text, q_f32, BF(q_bf16), inv_norm, B, D, ldq, ST(stream));
}

int spn_combine_l2norm_bwd(const float* q_f32, const float* inv_norm, const float* dq, float* dtext, int B, int D,
                           void* stream) {
    return combine_l2norm_bwd(q_f32, inv_norm, dq, dtext, B, D, ST(stream));
}

int spn_combine_l2norm_bwd_scaled(const float* q_f32, const float* inv_norm, const float* dq, const float* scale_dev,
                                  float* dtext, int B, int D, void* stream) {
    if (!q_f32 || !inv_norm || !dq || !dtext || !scale_dev) return SPN_ERR_ARG;
    return combine_l2norm_bwd(q_f32, inv_norm, dq, dtext, B, D, ST(stream), scale_dev);
}

static BankArgs make_bank(const void* q, int ldq, const void* bank, const int64_t* labels, int B, int M, int D,
                          int m_begin, float inv_tau) {
    BankArgs a;
    a.q = CBF(q); a.ldq = ldq; a.bank = CBF(bank); a.labels = labels;
    a.B = B; a.M = M; a.D = D; a.m_begin = m_begin; a.inv_tau = inv_tau;
    return a;
}

int spn_bank_stats_fwd(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B, int M, int D,
                       int m_begin, float inv_tau, float* stats, void* ws, size_t ws_bytes, void* stream) {
    if (!stats) return SPN_ERR_ARG;
    return bank_stats_fwd(make_bank(q_bf16, ldq, bank_bf16, labels, B, M, D, m_begin, inv_tau), stats, (float*)ws,
                          ws_bytes, ST(stream));
}

int spn_bank_loss_finalize(const float* stats, int nshards, int B, int64_t M_total, float label_smoothing,
                           float* row_lse, float* row_loss, float* loss_mean, void* stream) {
    if (!stats) return SPN_ERR_ARG;
    return bank_loss_finalize(stats, nshards, B, M_total, label_smoothing, row_lse, row_loss, loss_mean, ST(stream));
}

int spn_bank_grad_q(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B, int M, int D,
                    int m_begin, float inv_tau, const float* row_lse, float label_smoothing, int64_t M_total,
                    float grad_scale, float* dq, void* ws, size_t ws_bytes, void* stream) {
    return bank_grad_q(make_bank(q_bf16, ldq, bank_bf16, labels, B, M, D, m_begin, inv_tau), row_lse, label_smoothing,
                       M_total, grad_scale, dq, (float*)ws, ws_bytes, ST(stream));
}

size_t spn_bank_logits_bytes(int B, int M) { return bank_saved_bytes_any(B, M); }
int spn_bank_config(int mode) { return bank_config(mode); }

int spn_gemm_config(int key, int value) {
    if (key == 0) return gemm_nt_persist_set(value);
    return SPN_ERR_ARG;
}

int spn_bank_stats_fwd_save(const void* q_bf16, int ldq, const void* bank, const float* bank_scale, const int64_t* labels, int B,
                            int M, int D, int m_begin, float inv_tau, float* stats, float* logits_save, void* ws,
                            size_t ws_bytes, void* stream) {
    if (!stats || !logits_save) return SPN_ERR_ARG;
    BankArgs a = make_bank(q_bf16, ldq, bank, labels, B, M, D, m_begin, inv_tau);
    a.bank_scale = bank_scale;
    return bank_stats_fwd(a, stats, (float*)ws, ws_bytes, ST(stream), logits_save);
}

int spn_bank_grad_q_saved(const void* q_bf16, int ldq, const void* bank, const float* bank_scale, const int64_t* labels, int B,
                          int M, int D, int m_begin, float inv_tau, const float* logits_saved, const float* row_lse,
                          float label_smoothing, int64_t M_total, float grad_scale, float* dq, void* ws, size_t ws_bytes,
                          void* stream) {
    if (!logits_saved) return SPN_ERR_ARG;
    BankArgs a = make_bank(q_bf16, ldq, bank, labels, B, M, D, m_begin, inv_tau);
    a.bank_scale = bank_scale;
    return bank_grad_q(a, row_lse, label_smoothing, M_total, grad_scale, dq, (float*)ws, ws_bytes, ST(stream), logits_saved);
}

size_t spn_negtype_workspace_bytes(int B, int D) { return negtype_workspace_bytes(B, D); }

int spn_negtype_head(const float* refer, const float* text, const float* target, int B, int D, float inv_tau, int neg_type,
                     float* loss, float* d_refer, float* d_text, float* d_target, void* ws, size_t ws_bytes, void* stream) {
    return negtype_head(refer, text, target, B, D, inv_tau, neg_type, loss, d_refer, d_text, d_target, (float*)ws, ws_bytes,
                        ST(stream));
}

int spn_bank_step_ok(int B, int M, int D, int fp8) {
    BankArgs a = make_bank(nullptr, D, nullptr, nullptr, B, M, D, 0, 1.0f);
    static const float one = 1.0f;
    a.bank_scale = fp8 ? &one : nullptr;          // only tested against nullptr
    return bank_step_ok(a) ? 1 : 0;
}

int spn_bank_step(const void* q_bf16, int ldq, const void* bank, const float* bank_scale, const int64_t* labels, int B, int M,
                  int D, float inv_tau, float grad_scale, float* logits_save, float* row_lse, float* row_loss, float* loss_mean,
                  float* dq, void* stream) {
    BankArgs a = make_bank(q_bf16, ldq, bank, labels, B, M, D, 0, inv_tau);
    a.bank_scale = bank_scale;
    return bank_step(a, logits_save, grad_scale, row_lse, row_loss, loss_mean, dq, ST(stream));
}

size_t spn_bank_workspace_bytes(int B, int M, int D) { return bank_workspace_bytes(B, M, D); }
size_t spn_bank_workspace_bytes_fp8(int B, int M, int D) { return bank_workspace_bytes_fp8(B, M, D); }

int spn_bank_quantize_fp8(const float* bank, int M, int D, int Dp, void* bank_fp8, float* scale, void* stream) {
    return bank_quantize_fp8(bank, M, D, Dp, (uint8_t*)bank_fp8, scale, ST(stream));
}

int spn_bank_dequant_fp8(const void* bank_fp8, const float* bank_scale, int M, int D, void* bank_bf16, void* stream) {
    return bank_dequant_fp8((const uint8_t*)bank_fp8, bank_scale, M, D, (bf16_t*)bank_bf16, ST(stream));
}

int spn_bank_stats_fwd_fp8(const void* q_bf16, int ldq, const void* bank_fp8, const float* bank_scale,
                           const int64_t* labels, int B, int M, int D, int m_begin, float inv_tau, float* stats, void* ws,
                           size_t ws_bytes, void* stream) {
    if (!stats || !bank_scale) return SPN_ERR_ARG;
    BankArgs a = make_bank(q_bf16, ldq, bank_fp8, labels, B, M, D, m_begin, inv_tau);
    a.bank_scale = bank_scale;
    return bank_stats_fwd(a, stats, (float*)ws, ws_bytes, ST(stream));
}

int spn_bank_grad_q_fp8(const void* q_bf16, int ldq, const void* bank_fp8, const float* bank_scale, const int64_t* labels,
                        int B, int M, int D, int m_begin, float inv_tau, const float* row_lse, float label_smoothing,
                        int64_t M_total, float grad_scale, float* dq, void* ws, size_t ws_bytes, void* stream) {
    if (!bank_scale) return SPN_ERR_ARG;
    BankArgs a = make_bank(q_bf16, ldq, bank_fp8, labels, B, M, D, m_begin, inv_tau);
    a.bank_scale = bank_scale;
    return bank_grad_q(a, row_lse, label_smoothing, M_total, grad_scale, dq, (float*)ws, ws_bytes, ST(stream));
}

int spn_bank_stats_fwd_tokmax(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B,
                              int n_targets, int D, int t_begin, float inv_tau, float* stats, void* ws, size_t ws_bytes,
                              void* stream) {
    if (!stats || n_targets <= 0 || n_targets > (INT32_MAX >> 5)) return SPN_ERR_ARG;
    BankArgs a = make_bank(q_bf16, ldq, bank_bf16, labels, B, n_targets * 32, D, t_begin, inv_tau);
    a.group = 32;
    return bank_stats_fwd(a, stats, (float*)ws, ws_bytes, ST(stream));
}

int spn_bank_grad_q_tokmax(const void* q_bf16, int ldq, const void* bank_bf16, const int64_t* labels, int B, int n_targets,
                           int D, int t_begin, float inv_tau, const float* row_lse, float label_smoothing,
                           int64_t targets_total, float grad_scale, float* dq, void* ws, size_t ws_bytes, void* stream) {
    if (n_targets <= 0 || n_targets > (INT32_MAX >> 5)) return SPN_ERR_ARG;
    BankArgs a = make_bank(q_bf16, ldq, bank_bf16, labels, B, n_targets * 32, D, t_begin, inv_tau);
    a.group = 32;
    return bank_grad_q(a, row_lse, label_smoothing, targets_total, grad_scale, dq, (float*)ws, ws_bytes, ST(stream));
}

int spn_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float inv_scale, const float* found_inf, void* stream) {
    if (!p || !g || !m || !v) return SPN_ERR_ARG;
    return adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, inv_scale, found_inf, ST(stream));
}

int spn_adamw_step_scaled(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, const float* grad_scale, const float* found_inf,
                          void* stream) {
    if (!p || !g || !m || !v) return SPN_ERR_ARG;
    return adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, 1.0f, found_inf, ST(stream), grad_scale);
}

int spn_adamw_tick(float* step_dev, const float* found_inf, void* stream) { return adamw_tick(step_dev, found_inf, ST(stream)); }

int spn_adamw_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, const float* step_dev, const float* grad_scale, const float* found_inf,
                       void* stream) {
    if (!p || !g || !m || !v || !step_dev) return SPN_ERR_ARG;
    return adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, 0, 1.0f, found_inf, ST(stream), grad_scale, step_dev);
}


int spn_grad_check_finite(const float* g, size_t n, float* found_inf, void* stream) {
    if (!g || !found_inf) return SPN_ERR_ARG;
    return grad_unscale_check(const_cast<float*>(g), n, 1.0f, found_inf, ST(stream));
}

int spn_cosine_scores_f64(const float* q, const float* gallery, int Nq, int Ng, int D, double* scores, void* stream) {
    if (!q || !gallery || !scores) return SPN_ERR_ARG;
    return cosine_scores_f64(q, gallery, Nq, Ng, D, scores, ST(stream));
}

int spn_topk_from_scores(const double* scores, int Nq, int Ng, int K, const int32_t* exclude, int32_t* idx,
                         double* val, void* stream) {
    if (!scores || !idx) return SPN_ERR_ARG;
    return topk_from_scores(scores, Nq, Ng, K, exclude, idx, val, ST(stream));
}

static TextCfg tc(const spn_text_cfg* c) {
    TextCfg t;
    t.B = c->B; t.L = c->L; t.L_ctx = c->L_ctx; t.W = c->W; t.H = c->H; t.layers = c->layers; t.D = c->D;
    t.vocab = c->vocab; t.T = c->T; t.pool = c->pool;
    return t;
}

int spn_text_layout(const spn_text_cfg* cfg, spn_text_layout_t* out) {
    if (!cfg || !out) return SPN_ERR_ARG;
    text_layout(tc(cfg), reinterpret_cast<TextLayout*>(out));
    return SPN_OK;
}

size_t spn_text_act_bytes(const spn_text_cfg* cfg) { return cfg ? text_act_bytes(tc(cfg)) : 0; }
size_t spn_text_ws_bytes(const spn_text_cfg* cfg) { return cfg ? text_ws_bytes(tc(cfg)) : 0; }

int spn_text_refresh_bf16(const spn_text_cfg* cfg, const float* params, void* weights_bf16, void* stream) {
    if (!cfg || !params || !weights_bf16) return SPN_ERR_ARG;
    return text_refresh_bf16(tc(cfg), params, BF(weights_bf16), ST(stream));
}

int spn_text_fwd(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                 void* acts, float* feats, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !acts || !feats) return SPN_ERR_ARG;
    if (cfg->T != 0) return SPN_ERR_ARG;
    return text_fwd(tc(cfg), params, CBF(weights_bf16), ids, nullptr, (char*)acts, feats, ST(stream));
}

int spn_text_fwd_packed(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        const int32_t* cu_seqlens, void* acts, float* feats, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !cu_seqlens || !acts || !feats || cfg->T <= 0) return SPN_ERR_ARG;
    return text_fwd(tc(cfg), params, CBF(weights_bf16), ids, cu_seqlens, (char*)acts, feats, ST(stream));
}

int spn_text_bwd(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                 void* acts, const float* dfeats, float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !acts || !dfeats || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd(tc(cfg), params, CBF(weights_bf16), ids, (char*)acts, dfeats, grads, (char*)ws, ws_bytes,
                    ST(stream));
}

int spn_text_fwd_tokens(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        void* acts, float* feats, float* tokens, void* tokens_bf16, float* tok_mean, float* tok_rstd,
                        void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !acts || !feats || !tokens || !tok_mean || !tok_rstd) return SPN_ERR_ARG;
    return text_fwd_tokens(tc(cfg), params, CBF(weights_bf16), ids, (char*)acts, feats, tokens, BF(tokens_bf16), tok_mean,
                           tok_rstd, ST(stream));
}

int spn_text_bwd_tokens(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        void* acts, const float* dfeats, const float* dtokens, const float* tok_mean, const float* tok_rstd,
                        float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !acts || !dfeats || !dtokens || !tok_mean || !tok_rstd || !grads || !ws)
        return SPN_ERR_ARG;
    return text_bwd_tokens(tc(cfg), params, CBF(weights_bf16), ids, (char*)acts, dfeats, dtokens, tok_mean, tok_rstd, grads,
                           (char*)ws, ws_bytes, ST(stream));
}

int spn_text_bwd_tokens_head(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                             const float* dfeats, const float* dtokens, const float* tok_mean, const float* tok_rstd,
                             float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !acts || !dfeats || !dtokens || !tok_mean || !tok_rstd || !grads || !ws)
        return SPN_ERR_ARG;
    return text_bwd_tokens_head(tc(cfg), params, CBF(weights_bf16), (char*)acts, dfeats, dtokens, tok_mean, tok_rstd, grads,
                                (char*)ws, ws_bytes, ST(stream));
}

int spn_text_bwd_tail_tokens(const spn_text_cfg* cfg, const int32_t* ids, void* acts, float* grads, void* ws, size_t ws_bytes,
                             void* stream) {
    if (!cfg || !ids || !acts || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd_tail_tokens(tc(cfg), ids, (char*)acts, grads, (char*)ws, ws_bytes, ST(stream));
}

size_t spn_tg_ws_bytes(int B, int C) { return tg_ws_bytes(B, C); }

int spn_tg_tokenlearn_fwd(const float* z, const float* w, const float* bias, float* attn, float* mod_tokens, int B, int L,
                          int C, int S, int G, void* stream) {
    return tg_tokenlearn_fwd(z, w, bias, attn, mod_tokens, B, L, C, S, G, ST(stream));
}

int spn_tg_tokenlearn_bwd(const float* z, const float* w, const float* attn, const float* dmod_tokens, void* dz_bf16,
                          float* dw, float* dbias, void* ws, size_t ws_bytes, int B, int L, int C, int S, int G,
                          void* stream) {
    return tg_tokenlearn_bwd(z, w, attn, dmod_tokens, BF(dz_bf16), dw, dbias, (float*)ws, ws_bytes, B, L, C, S, G,
                             ST(stream));
}

int spn_tg_fuse_prep(const float* feats, const float* masks, const float* ref_tokens, float* mod_tokens, void* x_bf16,
                     float* x_f32, int B, int C, int S, int G, void* stream) {
    return tg_fuse_prep(feats, masks, ref_tokens, mod_tokens, BF(x_bf16), x_f32, B, C, S, G, ST(stream));
}

int spn_tg_img_finish(const float* feats, const float* masks, float* tokens, float* pooled, int B, int C, int S, int G,
                      void* stream) {
    return tg_img_finish(feats, masks, tokens, pooled, B, C, S, G, ST(stream));
}

int spn_tg_gate_fwd(const float* hpre, const float* w2, const float* b2, const float* ref_tokens, const float* mod_tokens,
                    float* remain, float* pooled, int B, int NT, int C, void* stream) {
    return tg_gate_fwd(hpre, w2, b2, ref_tokens, mod_tokens, remain, pooled, B, NT, C, ST(stream));
}

int spn_tg_gate_bwd(const float* dpooled, const float* ref_tokens, const float* mod_tokens, const float* remain,
                    const float* hpre, const float* w2, float* dmod_tokens, float* dh, float* dh_t, float* dw2, float* db1,
                    float* db2, void* ws, size_t ws_bytes, int B, int NT, int C, void* stream) {
    return tg_gate_bwd(dpooled, ref_tokens, mod_tokens, remain, hpre, w2, dmod_tokens, dh, dh_t, dw2, db1, db2, (float*)ws,
                       ws_bytes, B, NT, C, ST(stream));
}

int spn_tg_mod_bwd(const float* dx, float* dmod_tokens, const float* feats, const float* masks, float* dfeats, float* dmasks,
                   void* ws, size_t ws_bytes, int B, int C, int S, int G, void* stream) {
    return tg_mod_bwd(dx, dmod_tokens, feats, masks, dfeats, dmasks, (float*)ws, ws_bytes, B, C, S, G, ST(stream));
}

static_assert(sizeof(spn_vision_cfg) == sizeof(VisionCfg), "spn_vision_cfg layout");
static_assert(sizeof(spn_vision_layout_t) == sizeof(VisionLayout), "spn_vision_layout_t layout");

static VisionCfg vc(const spn_vision_cfg* c) {
    VisionCfg v;
    v.B = c->B; v.res = c->res; v.patch = c->patch; v.W = c->W; v.H = c->H; v.layers = c->layers; v.D = c->D;
    v.kind = c->kind;
    return v;
}

int spn_vision_layout(const spn_vision_cfg* cfg, spn_vision_layout_t* out) {
    if (!cfg || !out) return SPN_ERR_ARG;
    vision_layout(vc(cfg), reinterpret_cast<VisionLayout*>(out));
    return SPN_OK;
}

size_t spn_vision_ws_bytes(const spn_vision_cfg* cfg) { return cfg ? vision_ws_bytes(vc(cfg)) : 0; }

int spn_vision_refresh_bf16(const spn_vision_cfg* cfg, const float* params, void* weights_bf16, void* stream) {
    if (!cfg || !params || !weights_bf16) return SPN_ERR_ARG;
    return vision_refresh_bf16(vc(cfg), params, BF(weights_bf16), ST(stream));
}

int spn_vision_fwd(const spn_vision_cfg* cfg, const float* params, const void* weights_bf16, const float* image,
                   void* ws, size_t ws_bytes, float* feats, float* tokens_out, void* stream) {
    if (!cfg || !params || !weights_bf16 || !image || !ws || !feats) return SPN_ERR_ARG;
    return vision_fwd(vc(cfg), params, CBF(weights_bf16), image, (char*)ws, ws_bytes, feats, tokens_out, ST(stream));
}

size_t spn_vision_train_act_bytes(const spn_vision_cfg* cfg) { return cfg ? vision_train_act_bytes(vc(cfg)) : 0; }
size_t spn_vision_bwd_ws_bytes(const spn_vision_cfg* cfg) { return cfg ? vision_bwd_ws_bytes(vc(cfg)) : 0; }

int spn_vision_fwd_train(const spn_vision_cfg* cfg, const float* params, const void* weights_bf16, const float* image,
                         void* acts, float* feats, void* stream) {
    if (!cfg || !params || !weights_bf16 || !image || !acts || !feats) return SPN_ERR_ARG;
    return vision_fwd_train(vc(cfg), params, CBF(weights_bf16), image, (char*)acts, feats, ST(stream));
}

int spn_vision_bwd(const spn_vision_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                   const float* dfeats, float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !acts || !dfeats || !grads || !ws) return SPN_ERR_ARG;
    return vision_bwd(vc(cfg), params, CBF(weights_bf16), (char*)acts, dfeats, grads, (char*)ws, ws_bytes, ST(stream));
}

int spn_inbatch_grad_t(const void* q_bf16, const void* t_bf16, int ld, const float* row_lse, int B, int D,
                       float inv_tau, float grad_scale, float* dt, void* stream) {
    if (!q_bf16 || !t_bf16 || !row_lse || !dt) return SPN_ERR_ARG;
    return inbatch_grad_t(CBF(q_bf16), CBF(t_bf16), ld, row_lse, B, D, inv_tau, grad_scale, dt, ST(stream));
}

static_assert(sizeof(spn_fusion_cfg) == sizeof(FusionCfg), "spn_fusion_cfg layout");
static_assert(sizeof(spn_fusion_layout_t) == sizeof(FusionLayout), "spn_fusion_layout_t layout");

static FusionCfg fc(const spn_fusion_cfg* c) {
    FusionCfg f;
    f.B = c->B; f.L = c->L; f.S = c->S; f.W = c->W; f.H = c->H; f.layers = c->layers; f.I = c->I; f.E = c->E;
    f.Dp = c->Dp; f.vocab = c->vocab; f.max_pos = c->max_pos; f.T = c->T;
    return f;
}

int spn_fusion_layout(const spn_fusion_cfg* cfg, spn_fusion_layout_t* out) {
    if (!cfg || !out) return SPN_ERR_ARG;
    fusion_layout(fc(cfg), reinterpret_cast<FusionLayout*>(out));
    return SPN_OK;
}

// ------------------------------------------------------------------------- absorbed cross-attention (xattn.hip)
int spn_xattn_ok(int B, int L, int H, int S, int E) { return xattn_absorb_ok(B, L, H, S, E, H * 64) ? 1 : 0; }
int spn_xattn_sp(int S) { return xattn_sp(S); }

int spn_xattn_fwd(const void* q, const void* wkv, const void* wkv_t, const float* bkv, const void* x, const int32_t* cu, void* qa,
                  void* p, void* oa, void* ctx, int B, int L, int H, int S, int E, int T, float scale, void* stream) {
    if (!q || !wkv || !wkv_t || !bkv || !x || !qa || !p || !oa || !ctx) return SPN_ERR_ARG;
    const int W = H * 64;
    if (!xattn_absorb_ok(B, L, H, S, E, W)) return SPN_ERR_SHAPE;
    const int rows = cu ? T : B * L;
    if (rows <= 0 || (cu && (T < B || (int64_t)T > (int64_t)B * L))) return SPN_ERR_ARG;
    hipStream_t st = ST(stream);
    int rc = xattn_head_expand(CBF(q), W, CBF(wkv_t), 2 * W, 0, BF(qa), rows, H, E, scale, st);
    if (rc == SPN_OK) rc = xattn_scores_softmax(CBF(qa), CBF(x), BF(p), B, L * H, S, E, st, cu, H);
    if (rc == SPN_OK) rc = xattn_apply(CBF(p), CBF(x), BF(oa), B, L * H, S, E, st, cu, H, (int64_t)rows * H);
    if (rc == SPN_OK) rc = xattn_head_contract(CBF(oa), CBF(wkv), W, bkv, BF(ctx), W, rows, H, E, 1.0f, st);
    return rc;
}

int spn_xattn_bwd(const void* dctx, const void* ctx, const void* q, const void* wkv, const void* wkv_t, const float* bkv,
                  const void* x, const int32_t* cu, const void* p, const void* oa, void* doa, void* ds, void* dqa, float* delta,
                  void* dq, float* dwkv, float* dbkv, int B, int L, int H, int S, int E, int T, float scale, void* stream) {
    if (!dctx || !ctx || !q || !wkv || !wkv_t || !bkv || !x || !p || !oa || !doa || !ds || !dqa || !delta || !dq || !dwkv || !dbkv)
        return SPN_ERR_ARG;
    const int W = H * 64;
    if (!xattn_absorb_ok(B, L, H, S, E, W)) return SPN_ERR_SHAPE;
    const int rows = cu ? T : B * L, R = L * H;
    if (rows <= 0 || (cu && (T < B || (int64_t)T > (int64_t)B * L))) return SPN_ERR_ARG;
    hipStream_t st = ST(stream);
    int rc = xattn_delta(CBF(dctx), CBF(ctx), bkv + W, delta, rows, H, st);
    if (rc == SPN_OK) rc = xattn_head_expand(CBF(dctx), W, CBF(wkv_t), 2 * W, W, BF(doa), rows, H, E, 1.0f, st);
    if (rc == SPN_OK) rc = xattn_dscores(CBF(doa), CBF(x), CBF(p), delta, BF(ds), B, R, S, E, st, cu, H);
    if (rc == SPN_OK) rc = xattn_apply(CBF(ds), CBF(x), BF(dqa), B, R, S, E, st, cu, H, (int64_t)rows * H);
    if (rc == SPN_OK) rc = xattn_head_contract(CBF(dqa), CBF(wkv), 0, nullptr, BF(dq), W, rows, H, E, scale, st);
    if (rc == SPN_OK)
        rc = xattn_wgrad(CBF(q), 0, CBF(dqa), 0, CBF(dctx), 0, CBF(oa), 0, dwkv, dbkv, 0, 1, rows, W, H, E, scale, st);
    return rc;
}

int spn_fusion_packed_ok(const spn_fusion_cfg* cfg) { return cfg ? fusion_packed_ok(fc(cfg)) : 0; }
size_t spn_fusion_act_bytes(const spn_fusion_cfg* cfg) { return cfg ? fusion_act_bytes(fc(cfg)) : 0; }
size_t spn_fusion_ws_bytes(const spn_fusion_cfg* cfg) { return cfg ? fusion_ws_bytes(fc(cfg)) : 0; }

int spn_fusion_refresh_bf16(const spn_fusion_cfg* cfg, const float* params, void* weights_bf16, void* stream) {
    if (!cfg || !params || !weights_bf16) return SPN_ERR_ARG;
    return fusion_refresh_bf16(fc(cfg), params, BF(weights_bf16), ST(stream));
}

int spn_fusion_fwd(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                   const int32_t* mask, const float* enc, void* acts, float* proj_out, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !enc || !acts || !proj_out) return SPN_ERR_ARG;
    return fusion_fwd(fc(cfg), params, CBF(weights_bf16), ids, mask, enc, (char*)acts, proj_out, ST(stream));
}

int spn_fusion_fwd_bank(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                        const int32_t* mask, const void* token_bank_bf16, int64_t bank_rows, const int64_t* token_idx, void* acts,
                        float* proj_out, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !token_bank_bf16 || !token_idx || bank_rows <= 0 || !acts || !proj_out)
        return SPN_ERR_ARG;
    return fusion_fwd(fc(cfg), params, CBF(weights_bf16), ids, mask, nullptr, (char*)acts, proj_out, ST(stream), CBF(token_bank_bf16),
                      token_idx, bank_rows);
}

int spn_cast_bf16_f32(const void* x_bf16, float* y, size_t n, void* stream) { return cast_bf16_f32(CBF(x_bf16), y, n, ST(stream)); }

int spn_sum_ranks_bf16(const void* chunks_bf16, int n_ranks, size_t m, void* out_bf16, void* stream) {
    return sum_ranks_bf16(CBF(chunks_bf16), n_ranks, m, BF(out_bf16), ST(stream));
}

int spn_jpeg_decode_batch(const uint8_t* bytes, const spn_jpeg_image* images, int n_images, const spn_jpeg_segment* segments,
                          int n_segments, const spn_jpeg_huff* huff, const uint16_t* qtabs, int16_t* coefs, size_t coef_elems,
                          uint8_t* planes, uint8_t* rgb, int max_blocks, int max_pixels, void* stream) {
    return jpeg_decode_batch(bytes, images, n_images, segments, n_segments, huff, qtabs, coefs, coef_elems, planes, rgb, max_blocks,
                             max_pixels, ST(stream));
}

int spn_sum_ranks_f32(const float* chunks, int n_ranks, size_t m, float* out, void* stream) {
    return sum_ranks_f32(chunks, n_ranks, m, out, ST(stream));
}

int spn_gather_bank_rows_bf16(const void* bank_bf16, int64_t bank_rows, const int64_t* idx, void* out_bf16, int B, int64_t row_elems,
                              void* stream) {
    if (row_elems <= 0) return SPN_ERR_ARG;
    return gather_bank_rows_bf16(CBF(bank_bf16), idx, bank_rows, BF(out_bf16), B, (size_t)row_elems, ST(stream));
}

int spn_tau_grad(const float* q, const float* dqk, int lddq, const float* tau_dev, int B, int D, float alpha, const float* scale_dev,
                 float* dtau, float* inv_tau, void* stream) {
    return tau_grad(q, dqk, lddq, tau_dev, B, D, alpha, scale_dev, dtau, inv_tau, ST(stream));
}

int spn_fusion_bwd(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                   void* acts, const float* dproj, float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !acts || !dproj || !grads || !ws) return SPN_ERR_ARG;
    return fusion_bwd(fc(cfg), params, CBF(weights_bf16), ids, (char*)acts, dproj, grads, (char*)ws, ws_bytes, ST(stream));
}

int spn_fusion_bwd_phase(const spn_fusion_cfg* cfg, const float* params, const void* weights_bf16, const int32_t* ids,
                         void* acts, const float* dproj, float* grads, void* ws, size_t ws_bytes, int phase, int l_lo, int l_hi,
                         void* stream) {
    if (!cfg || !params || !weights_bf16 || !ids || !acts || !grads || !ws) return SPN_ERR_ARG;
    return fusion_bwd_phase(fc(cfg), params, CBF(weights_bf16), ids, (char*)acts, dproj, grads, (char*)ws, ws_bytes, phase, l_lo,
                            l_hi, ST(stream));
}

int spn_scale_cast_bf16(const float* x, const float* scale_dev, int reciprocal, void* out_bf16, int B, int D, int ldo,
                        void* stream) {
    if (!x || !scale_dev || !out_bf16) return SPN_ERR_ARG;
    return scale_cast_bf16(x, scale_dev, reciprocal, BF(out_bf16), B, D, ldo, ST(stream));
}

int spn_text_bwd_head(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                      const float* dfeats, float* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !acts || !dfeats || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd_head(tc(cfg), params, CBF(weights_bf16), (char*)acts, dfeats, grads, (char*)ws, ws_bytes, ST(stream));
}

int spn_text_bwd_layer(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                       float* grads, int layer, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !acts || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd_layer(tc(cfg), params, CBF(weights_bf16), (char*)acts, grads, layer, (char*)ws, ws_bytes, ST(stream));
}

int spn_text_bwd_layer_deferred(const spn_text_cfg* cfg, const float* params, const void* weights_bf16, void* acts,
                                float* grads, int layer, void* ws, size_t ws_bytes, void* stream) {
    if (!cfg || !params || !weights_bf16 || !acts || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd_layer_deferred(tc(cfg), params, CBF(weights_bf16), (char*)acts, grads, layer, (char*)ws, ws_bytes,
                                   ST(stream));
}

int spn_text_bwd_wgrad(const spn_text_cfg* cfg, void* acts, float* grads, int layer_begin, int layer_end, void* ws,
                       size_t ws_bytes, void* stream) {
    if (!cfg || !acts || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd_wgrad(tc(cfg), (char*)acts, grads, layer_begin, layer_end, (char*)ws, ws_bytes, ST(stream));
}

int spn_text_bwd_tail(const spn_text_cfg* cfg, const int32_t* ids, void* acts, float* grads, void* ws,
                      size_t ws_bytes, void* stream) {
    if (!cfg || !ids || !acts || !grads || !ws) return SPN_ERR_ARG;
    return text_bwd_tail(tc(cfg), ids, (char*)acts, grads, (char*)ws, ws_bytes, ST(stream));
}

int spn_preprocess_image(const uint8_t* src_rgb, int H, int W, int pad_x, int pad_y, const int32_t* kx, const int32_t* bx,
                         int ksize_x, const int32_t* ky, const int32_t* by, int ksize_y, int crop_left, int crop_top, int dim,
                         const float* mean3, const float* std3, void* tmp, float* out_chw, uint8_t* out_u8_hwc,
                         void* stream) {
    return preprocess_image(src_rgb, H, W, pad_x, pad_y, kx, bx, ksize_x, ky, by, ksize_y, crop_left, crop_top, dim, mean3,
                            std3, (uint8_t*)tmp, out_chw, out_u8_hwc, ST(stream));
}

size_t spn_text_exact_ws_bytes(const spn_text_cfg* cfg) { return cfg ? text_exact_ws_bytes(tc(cfg)) : 0; }
size_t spn_vision_exact_ws_bytes(const spn_vision_cfg* cfg) { return cfg ? vision_exact_ws_bytes(vc(cfg)) : 0; }

int spn_text_fwd_exact(const spn_text_cfg* cfg, const float* params, const int32_t* ids, void* ws, size_t ws_bytes,
                       float* feats, void* stream) {
    if (!cfg || !params || !ids || !ws || !feats) return SPN_ERR_ARG;
    return text_fwd_exact(tc(cfg), params, ids, (char*)ws, ws_bytes, feats, ST(stream));
}

int spn_vision_fwd_exact(const spn_vision_cfg* cfg, const float* params, const float* image, void* ws, size_t ws_bytes,
                         float* feats, void* stream) {
    if (!cfg || !params || !image || !ws || !feats) return SPN_ERR_ARG;
    return vision_fwd_exact(vc(cfg), params, image, (char*)ws, ws_bytes, feats, ST(stream));
}

int spn_gemm_f32(const float* A, const float* B, int M, int N, int K, int lda, int ldb, int b_is_kn, const float* bias,
                 int act, const float* resid, int ldr, float* C, int ldc, float alpha, void* stream) {
    return gemm_f32(A, B, M, N, K, lda, ldb, b_is_kn, bias, act, resid, ldr, C, ldc, alpha, ST(stream));
}

int spn_im2col3x3_f32(const float* x, float* out, int B, int H, int W, int C, int stride, int nchw, int ldk, void* stream) {
    if (!x || !out) return SPN_ERR_ARG;
    return im2col3x3_f32(x, out, B, H, W, C, stride, nchw, ldk, ST(stream));
}

int spn_avgpool_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, int k, void* stream) {
    if (!x || !y) return SPN_ERR_ARG;
    return avgpool_nhwc_f32(x, y, B, H, W, C, k, ST(stream));
}

int spn_im2col3x3_nhwc_bf16(const void* x, void* out, int B, int H, int W, int Cp, int stride, void* stream) {
    return im2col3x3_nhwc_bf16((const bf16_t*)x, (bf16_t*)out, B, H, W, Cp, stride, ST(stream));
}

int spn_im2col3x3_stem_bf16(const float* image, void* out, int B, int H, int W, int stride, void* stream) {
    return im2col3x3_stem_bf16(image, (bf16_t*)out, B, H, W, stride, ST(stream));
}

int spn_relu_add_bf16(void* y, const void* resid, size_t n, void* stream) {
    return relu_add_bf16((bf16_t*)y, (const bf16_t*)resid, n, ST(stream));
}

int spn_avgpool_nhwc_bf16(const void* x, void* y, int B, int H, int W, int Cp, int k, void* stream) {
    return avgpool_nhwc_bf16((const bf16_t*)x, (bf16_t*)y, B, H, W, Cp, k, ST(stream));
}

int spn_attnpool_tokens_f32(const float* x, const float* pos, float* tok, int B, int HW, int C, void* stream) {
    if (!x || !pos || !tok) return SPN_ERR_ARG;
    return attnpool_tokens_f32(x, pos, tok, B, HW, C, ST(stream));
}

int spn_attnpool_attend_f32(const float* q, const float* k, const float* v, float* out, int B, int S, int H, void* stream) {
    if (!q || !k || !v || !out) return SPN_ERR_ARG;
    return attnpool_attend_f32(q, k, v, out, B, S, H, ST(stream));
}

}  // extern "C"
