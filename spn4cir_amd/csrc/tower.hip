// Transformer tower orchestration: the CLIP pre-LN residual block (clip/model.py:171-192)
// forward + backward, and the text tower around it (CLIP.encode_text, clip/model.py:345-358).
// Pure launch sequencing on one stream; every device op is a kernel from this library.
#include "tower.h"

namespace spn {

#define SPN_TRY(x)                  \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != SPN_OK) return rc__; \
    } while (0)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------ layouts
void block_param_offsets(int W, int64_t off[13]) {
    const int64_t w = W;
    int64_t o = 0;
    off[0] = o; o += w;            // ln1_g
    off[1] = o; o += w;            // ln1_b
    off[2] = o; o += 3 * w * w;    // w_qkv [3W, W]
    off[3] = o; o += 3 * w;        // b_qkv
    off[4] = o; o += w * w;        // w_o [W, W]
    off[5] = o; o += w;            // b_o
    off[6] = o; o += w;            // ln2_g
    off[7] = o; o += w;            // ln2_b
    off[8] = o; o += 4 * w * w;    // w_fc [4W, W]
    off[9] = o; o += 4 * w;        // b_fc
    off[10] = o; o += 4 * w * w;   // w_proj [W, 4W]
    off[11] = o; o += w;           // b_proj
    off[12] = o;                   // block size
}

int64_t block_bf16_size(int W) { return 24ll * W * W; }

BlockParams block_params_at(const float* p, const bf16_t* wb, int W) {
    int64_t o[13];
    block_param_offsets(W, o);
    const int64_t w2 = (int64_t)W * W;
    BlockParams P;
    P.ln1_g = p + o[0]; P.ln1_b = p + o[1]; P.b_qkv = p + o[3]; P.b_o = p + o[5];
    P.ln2_g = p + o[6]; P.ln2_b = p + o[7]; P.b_fc = p + o[9]; P.b_proj = p + o[11];
    P.w_qkv = wb;                 P.w_qkv_t = wb + 3 * w2;
    P.w_o = wb + 6 * w2;          P.w_o_t = wb + 7 * w2;
    P.w_fc = wb + 8 * w2;         P.w_fc_t = wb + 12 * w2;
    P.w_proj = wb + 16 * w2;      P.w_proj_t = wb + 20 * w2;
    return P;
}

BlockGrads block_grads_at(float* g, int W) {
    int64_t o[13];
    block_param_offsets(W, o);
    BlockGrads G;
    G.ln1_g = g + o[0]; G.ln1_b = g + o[1]; G.w_qkv = g + o[2]; G.b_qkv = g + o[3];
    G.w_o = g + o[4]; G.b_o = g + o[5]; G.ln2_g = g + o[6]; G.ln2_b = g + o[7];
    G.w_fc = g + o[8]; G.b_fc = g + o[9]; G.w_proj = g + o[10]; G.b_proj = g + o[11];
    return G;
}

// bf16 copies of one block's GEMM weights (+ their transposes for the backward-data GEMMs)
int block_refresh_bf16(const float* p, bf16_t* wb, int W, hipStream_t st) {
    int64_t o[13];
    block_param_offsets(W, o);
    const int64_t w2 = (int64_t)W * W;
    SPN_TRY(cast_transpose_f32_bf16(p + o[2], wb, wb + 3 * w2, 3 * W, W, st));
    SPN_TRY(cast_transpose_f32_bf16(p + o[4], wb + 6 * w2, wb + 7 * w2, W, W, st));
    SPN_TRY(cast_transpose_f32_bf16(p + o[8], wb + 8 * w2, wb + 12 * w2, 4 * W, W, st));
    SPN_TRY(cast_transpose_f32_bf16(p + o[10], wb + 16 * w2, wb + 20 * w2, W, 4 * W, st));
    return SPN_OK;
}

// all `layers` blocks in ONE launch (p / wb point at block 0, strides in elements)
int blocks_refresh_bf16(const float* p, int64_t p_stride, bf16_t* wb, int64_t wb_stride, int layers, int W, hipStream_t st) {
    int64_t o[13];
    block_param_offsets(W, o);
    const int64_t w2 = (int64_t)W * W;
    CastTransposeSet d;
    const int64_t src[4] = {o[2], o[4], o[8], o[10]};
    const int64_t dst[4] = {0, 6 * w2, 8 * w2, 16 * w2}, dst_t[4] = {3 * w2, 7 * w2, 12 * w2, 20 * w2};
    const int rows[4] = {3 * W, W, 4 * W, W}, cols[4] = {W, W, W, 4 * W};
    d.tile_start[0] = 0;
    for (int i = 0; i < 4; ++i) {
        d.src[i] = src[i]; d.dst[i] = dst[i]; d.dst_t[i] = dst_t[i]; d.rows[i] = rows[i]; d.cols[i] = cols[i];
        d.tile_start[i + 1] = d.tile_start[i] + ((rows[i] + 31) / 32) * ((cols[i] + 31) / 32);
    }
    d.p_stride = p_stride; d.wb_stride = wb_stride;
    return cast_transpose_multi(p, wb, d, layers, st);
}

size_t block_act_bytes(const BlockCfg& c) {
    const size_t T = (size_t)c.rows(), W = c.W;
    size_t b = 0;
    b += align256(T * W * 4);                 // x_in
    b += 2 * align256(T * 4);                 // mean1, rstd1
    b += align256(T * W * 2);                 // h1
    b += align256(T * 3 * W * 2);             // qkv
    b += align256((size_t)c.B * c.H * c.L * 4);   // lse
    b += align256(T * W * 2);                 // attn
    b += align256(T * W * 4);                 // x_mid
    b += 2 * align256(T * 4);                 // mean2, rstd2
    b += align256(T * W * 2);                 // h2
    b += 2 * align256(T * 4 * W * 2);         // pre, u
    return b;
}

BlockActs block_acts_at(char* base, const BlockCfg& c) {
    const size_t T = (size_t)c.rows(), W = c.W;
    BlockActs A;
    char* p = base;
    auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
    A.x_in = (float*)take(T * W * 4);
    A.mean1 = (float*)take(T * 4);
    A.rstd1 = (float*)take(T * 4);
    A.h1 = (bf16_t*)take(T * W * 2);
    A.qkv = (bf16_t*)take(T * 3 * W * 2);
    A.lse = (float*)take((size_t)c.B * c.H * c.L * 4);
    A.attn = (bf16_t*)take(T * W * 2);
    A.x_mid = (float*)take(T * W * 4);
    A.mean2 = (float*)take(T * 4);
    A.rstd2 = (float*)take(T * 4);
    A.h2 = (bf16_t*)take(T * W * 2);
    A.pre = (bf16_t*)take(T * 4 * W * 2);
    A.u = (bf16_t*)take(T * 4 * W * 2);
    A.x_out = nullptr;
    return A;
}

// scratch for one block's backward + the op workspaces (shared, used sequentially)
size_t block_bwd_scratch_bytes(const BlockCfg& c) {
    const size_t T = (size_t)c.rows(), W = c.W;
    size_t b = 0;
    b += align256(T * 4 * W * 2);   // dpre
    b += align256(T * W * 2);       // dh
    b += align256(T * W * 2);       // dattn
    b += align256(T * 3 * W * 2);   // dqkv
    b += align256((size_t)c.B * c.H * c.L * 4);   // delta
    return b;
}

size_t block_op_ws_bytes(const BlockCfg& c) {
    const int T = c.rows(), W = c.W;
    size_t m = 0;
    auto mx = [&](size_t v) { if (v > m) m = v; };
    mx(gemm_tn_workspace_bytes(T, W, 4 * W));
    mx(gemm_tn_workspace_bytes(T, 4 * W, W));
    mx(gemm_tn_workspace_bytes(T, W, W));
    mx(gemm_tn_workspace_bytes(T, 3 * W, W));
    mx(gemm_tn2_pair_workspace_bytes(T, 3 * W, W, W, W));
    mx(gemm_tn_grouped_workspace_bytes(T));
    mx(colsum_workspace_bytes(T, 4 * W));
    mx(layernorm_bwd_workspace_bytes(T, W));
    return align256(m);
}

// SPN_FUSE_RESID=0 (A/B switch): the text tower's residual GEMMs keep their fp32 read-add-write epilogue (block_fwd)
static int fuse_resid_on() {
    static const int on = [] {
        const char* e = spn_env("SPN_FUSE_RESID");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on;
}

// SPN_AUX_GRAD=0 (A/B switch): keep the MLP's pre-activation and evaluate the activation's derivative in the backward GEMM
static int aux_grad_on() {
    static const int on = [] {
        const char* e = spn_env("SPN_AUX_GRAD");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    return on;
}

// -------------------------------------------------------------------------------- block
// Deferred residual adds (c.fuse_resid, the text tower's default): the two residual GEMMs of a block (out-projection,
// c_proj) store their result as bf16 - under the reference's autocast these Linear outputs are fp16 tensors too - and the
// add to the fp32 residual stream happens in the LayerNorm that follows (layernorm_fwd_add), which reads the stream anyway.
// A GEMM_RESID epilogue reads and writes 2 x 60 MB of fp32 at [19 712, 768] in one synchronised burst of all 231 tiles
// (out-projection 50 us against 26 us for the same product with a bf16 store, c_proj 103 against 71); the fused
// LayerNorm grows from 17 to ~33 us.  The bf16 results are parked in buffers that are dead at that point and that the
// consuming LayerNorm overwrites in place: the out-projection's in A.h2, c_proj's in the NEXT block's h1 (y_out).
//   x_prev != nullptr: A.h1 holds the previous block's c_proj result; x_in = x_prev + it is formed here.
//   y_out  != nullptr: c_proj stores bf16 there instead of x_out = x_mid + result.
// first half of a block: ln_1 (+ the previous block's deferred residual), the qkv projection and the attention - everything
// that mixes rows; what follows it (out-projection, ln_2, MLP) is row-wise
static int block_fwd_attn(const BlockCfg& c, const BlockParams& P, const BlockActs& A, hipStream_t st, const float* x_prev) {
    const int T = c.rows(), W = c.W;
    if (x_prev && !c.fuse_resid) return SPN_ERR_ARG;
    if (x_prev) SPN_TRY(layernorm_fwd_add(x_prev, A.h1, P.ln1_g, P.ln1_b, A.x_in, A.h1, A.mean1, A.rstd1, T, W, c.eps, st));
    else SPN_TRY(layernorm_fwd(A.x_in, P.ln1_g, P.ln1_b, A.h1, nullptr, A.mean1, A.rstd1, T, W, c.eps, st));
    {
        GemmEpilogue e;
        e.bias = P.b_qkv; e.out_bf16 = A.qkv; e.ldc = 3 * W;
        SPN_TRY(gemm_nt(A.h1, P.w_qkv, T, 3 * W, W, W, W, GEMM_STORE, e, st));
    }
    {
        AttnArgs a;
        a.q = A.qkv; a.k = A.qkv + W; a.v = A.qkv + 2 * W;
        a.ldq = a.ldk = a.ldv = 3 * W;
        a.o = A.attn; a.ldo = W; a.lse = A.lse; a.key_bias = nullptr;
        a.B = c.B; a.H = c.H; a.Lq = c.L; a.Lk = c.L; a.causal = c.causal; a.cu = c.cu;
        a.scale = 0.125f;
        SPN_TRY(attention_fwd(a, st));
    }
    return SPN_OK;
}

int block_fwd(const BlockCfg& c, const BlockParams& P, const BlockActs& A, hipStream_t st, const float* x_prev, bf16_t* y_out) {
    const int T = c.rows(), W = c.W;
    if ((x_prev || y_out) && !c.fuse_resid) return SPN_ERR_ARG;
    SPN_TRY(block_fwd_attn(c, P, A, st, x_prev));
    if (c.fuse_resid) {
        GemmEpilogue e;
        e.bias = P.b_o; e.out_bf16 = A.h2; e.ldc = W;
        SPN_TRY(gemm_nt(A.attn, P.w_o, T, W, W, W, W, GEMM_STORE, e, st));
        SPN_TRY(layernorm_fwd_add(A.x_in, A.h2, P.ln2_g, P.ln2_b, A.x_mid, A.h2, A.mean2, A.rstd2, T, W, c.eps, st));
    } else {
        GemmEpilogue e;
        e.bias = P.b_o; e.resid = A.x_in; e.ldr = W; e.out_f32 = A.x_mid; e.ldc = W;
        SPN_TRY(gemm_nt(A.attn, P.w_o, T, W, W, W, W, GEMM_RESID, e, st));
        SPN_TRY(layernorm_fwd(A.x_mid, P.ln2_g, P.ln2_b, A.h2, nullptr, A.mean2, A.rstd2, T, W, c.eps, st));
    }
    {
        GemmEpilogue e;
        e.bias = P.b_fc; e.act = c.act; e.aux_out = A.pre; e.aux_grad = aux_grad_on(); e.out_bf16 = A.u; e.ldc = 4 * W;
        SPN_TRY(gemm_nt(A.h2, P.w_fc, T, 4 * W, W, W, W, GEMM_STORE, e, st));
    }
    if (y_out) {
        GemmEpilogue e;
        e.bias = P.b_proj; e.out_bf16 = y_out; e.ldc = W;
        SPN_TRY(gemm_nt(A.u, P.w_proj, T, W, 4 * W, 4 * W, 4 * W, GEMM_STORE, e, st));
    } else {
        GemmEpilogue e;
        e.bias = P.b_proj; e.resid = A.x_mid; e.ldr = W; e.out_f32 = A.x_out; e.ldc = W;
        SPN_TRY(gemm_nt(A.u, P.w_proj, T, W, 4 * W, 4 * W, 4 * W, GEMM_RESID, e, st));
    }
    return SPN_OK;
}

// dx / dx_bf16: gradient w.r.t. the block output on entry, w.r.t. the block input on exit.
// SPN_TN_GROUP=0 (A/B switch): the four weight gradients of a block as separate split-K launches again
static bool tn_group_on() {
    static const bool on = [] {
        const char* e = spn_env("SPN_TN_GROUP");
        return !(e && e[0] == '0');
    }();
    return on;
}

int block_bwd(const BlockCfg& c, const BlockParams& P, const BlockActs& A, const BlockGrads& G, float* dx,
              bf16_t* dx_bf16, char* scratch, float* ws, size_t ws_bytes, hipStream_t st, const BwdOverlap* ov,
              bf16_t* dxb_group, const BwdDefer* defer) {
    const int T = c.rows(), W = c.W;
    const size_t Ts = (size_t)T;
    if (defer && ov) return SPN_ERR_ARG;
    char* p = scratch;
    auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
    bf16_t* dpre = (bf16_t*)take(Ts * 4 * W * 2);
    bf16_t* dh = (bf16_t*)take(Ts * W * 2);
    bf16_t* dattn = (bf16_t*)take(Ts * W * 2);
    bf16_t* dqkv = (bf16_t*)take(Ts * 3 * W * 2);
    float* delta = (float*)take((size_t)c.B * c.H * c.L * 4);
    if (defer) { dpre = defer->dpre; dqkv = defer->dqkv; }

    // weight-gradient GEMMs: side stream + own workspace when overlapping, otherwise in line
    const hipStream_t sw = ov ? ov->side : st;
    float* wws = ov ? ov->ws2 : ws;
    const size_t wws_bytes = ov ? ov->ws2_bytes : ws_bytes;
    // Grouped mode (dxb_group = a second [T, W] bf16 buffer): the block's four weight gradients dW = dY^T X feed nothing
    // downstream, so they are computed at the END of the block by ONE launch without per-problem split-K
    // (gemm_tn_grouped): 108 tiles x 2 slices instead of 3 launches of 36 tiles x 7 slices + 4 reductions.  The
    // gradient entering the block (dx_bf16) then has to survive until that launch, so the mid-block gradient goes to
    // dxb_group instead of overwriting it.
    const bool grouped = defer != nullptr || (!ov && dxb_group != nullptr && tn_group_on());
    bf16_t* dxb_mid = ov ? ov->dxb_alt : (defer ? defer->dx_mid : (grouped ? dxb_group : dx_bf16));   // between the two halves
    auto fork = [&](int i) -> int {                    // side stream may start once the main stream got here
        if (!ov) return SPN_OK;
        hipError_t e = hipEventRecord(ov->ev[i], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(ov->side, ov->ev[i], 0);
        return e == hipSuccess ? SPN_OK : (int)e;
    };
    auto join = [&](int i) -> int {                    // main stream waits for everything enqueued on the side stream
        if (!ov) return SPN_OK;
        hipError_t e = hipEventRecord(ov->ev[i], ov->side);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ov->ev[i], 0);
        return e == hipSuccess ? SPN_OK : (int)e;
    };

    // MLP
    SPN_TRY(fork(0));                                  // dx_bf16 is final
    {
        GemmEpilogue e;
        e.aux_in = A.pre; e.aux_grad = aux_grad_on(); e.act = c.act; e.out_bf16 = dpre; e.ldc = 4 * W;
        SPN_TRY(gemm_nt(dx_bf16, P.w_proj_t, T, 4 * W, W, W, W, GEMM_DACT, e, st));
    }
    if (!grouped) SPN_TRY(gemm_tn(dx_bf16, A.u, T, W, 4 * W, W, 4 * W, G.w_proj, 4 * W, 1.0f, 0, G.b_proj, wws, wws_bytes, sw));
    if (ov) {                                          // ev[4]: the side stream no longer reads dx_bf16
        hipError_t e = hipEventRecord(ov->ev[4], ov->side);
        if (e != hipSuccess) return (int)e;
    }
    SPN_TRY(fork(1));                                  // dpre is final
    {
        GemmEpilogue e;
        e.out_bf16 = dh; e.ldc = W;
        SPN_TRY(gemm_nt(dpre, P.w_fc_t, T, W, 4 * W, 4 * W, 4 * W, GEMM_STORE, e, st));
    }
    if (!grouped) SPN_TRY(gemm_tn(dpre, A.h2, T, 4 * W, W, 4 * W, W, G.w_fc, W, 1.0f, 0, G.b_fc, wws, wws_bytes, sw));
    const size_t lnp = layernorm_bwd_workspace_bytes(T, W);       // deferred: partials stay in the block's own buffer
    if (defer) SPN_TRY(layernorm_bwd(dh, nullptr, A.x_mid, P.ln2_g, A.mean2, A.rstd2, dx, 1, dxb_mid, G.ln2_g, G.ln2_b, 2, T, W,
                                     defer->ln_partials, lnp, st));
    else SPN_TRY(layernorm_bwd(dh, nullptr, A.x_mid, P.ln2_g, A.mean2, A.rstd2, dx, 1, dxb_mid, G.ln2_g, G.ln2_b, 0, T, W, ws,
                               ws_bytes, st));
    // attention
    SPN_TRY(fork(2));                                  // dxb_mid is final
    {
        GemmEpilogue e;
        e.out_bf16 = dattn; e.ldc = W;
        SPN_TRY(gemm_nt(dxb_mid, P.w_o_t, T, W, W, W, W, GEMM_STORE, e, st));
    }
    // the out-projection's weight gradient (W x W: 9 tiles) rides along with the qkv one below when they can share a launch
    const bool pair = !ov && !grouped && gemm_tn2_pair_ok(3 * W, W);
    if (!pair && !grouped) SPN_TRY(gemm_tn(dxb_mid, A.attn, T, W, W, W, W, G.w_o, W, 1.0f, 0, G.b_o, wws, wws_bytes, sw));
    {
        AttnBwdArgs g;
        AttnArgs& a = g.f;
        a.q = A.qkv; a.k = A.qkv + W; a.v = A.qkv + 2 * W;
        a.ldq = a.ldk = a.ldv = 3 * W;
        a.o = A.attn; a.ldo = W; a.lse = A.lse; a.key_bias = nullptr;
        a.B = c.B; a.H = c.H; a.Lq = c.L; a.Lk = c.L; a.causal = c.causal; a.cu = c.cu;
        a.scale = 0.125f;
        g.d_o = dattn; g.lddo = W;
        g.dq = dqkv; g.dk = dqkv + W; g.dv = dqkv + 2 * W;
        g.lddq = g.lddk = g.lddv = 3 * W;
        g.delta = delta;
        SPN_TRY(attention_bwd(g, st));
    }
    SPN_TRY(fork(3));                                  // dqkv is final
    {
        GemmEpilogue e;
        e.out_bf16 = dh; e.ldc = W;
        SPN_TRY(gemm_nt(dqkv, P.w_qkv_t, T, W, 3 * W, 3 * W, 3 * W, GEMM_STORE, e, st));
    }
    if (pair)
        SPN_TRY(gemm_tn2_pair(dqkv, A.h1, 3 * W, W, 3 * W, W, G.w_qkv, W, G.b_qkv, dxb_mid, A.attn, W, W, W, W, G.w_o, W, G.b_o,
                              T, wws, wws_bytes, sw));
    else if (!grouped) SPN_TRY(gemm_tn(dqkv, A.h1, T, 3 * W, W, 3 * W, W, G.w_qkv, W, 1.0f, 0, G.b_qkv, wws, wws_bytes, sw));
    if (grouped) {
        TnProblem q[4];
        q[0] = TnProblem{dx_bf16, A.u, G.w_proj, G.b_proj, W, 4 * W, W, 4 * W, 4 * W};
        q[1] = TnProblem{dpre, A.h2, G.w_fc, G.b_fc, 4 * W, W, 4 * W, W, W};
        q[2] = TnProblem{dqkv, A.h1, G.w_qkv, G.b_qkv, 3 * W, W, 3 * W, W, W};
        q[3] = TnProblem{dxb_mid, A.attn, G.w_o, G.b_o, W, W, W, W, W};
        if (defer) {
            for (int i = 0; i < 4; ++i) defer->problems[i] = q[i];
        } else {
            SPN_TRY(gemm_tn_grouped(q, 4, T, ws, ws_bytes, st));
        }
    }
    if (ov) {                                          // LayerNorm backward rewrites dx_bf16: w_proj's GEMM must be done with it
        hipError_t e = hipStreamWaitEvent(st, ov->ev[4], 0);
        if (e != hipSuccess) return (int)e;
    }
    if (defer) SPN_TRY(layernorm_bwd(dh, nullptr, A.x_in, P.ln1_g, A.mean1, A.rstd1, dx, 1, defer->dx_out, G.ln1_g, G.ln1_b, 2,
                                     T, W, (float*)((char*)defer->ln_partials + lnp), lnp, st));
    else SPN_TRY(layernorm_bwd(dh, nullptr, A.x_in, P.ln1_g, A.mean1, A.rstd1, dx, 1, dx_bf16, G.ln1_g, G.ln1_b, 0, T, W, ws,
                               ws_bytes, st));
    // every parameter gradient of the block is final in `st` order on return (DDP bucket hooks rely on it), and
    // the scratch / dxb_alt buffers are free for the next block
    SPN_TRY(join(5));
    return SPN_OK;
}

// --------------------------------------------------------------------------- text tower
static BlockCfg text_block_cfg(const TextCfg& c) {
    BlockCfg b;
    b.B = c.B; b.L = c.L; b.W = c.W; b.H = c.H; b.causal = 1; b.act = ACT_QUICKGELU; b.eps = 1e-5f;
    b.T = c.T;   // b.cu is filled in by the callers from the activation arena
    return b;
}

void text_layout(const TextCfg& c, TextLayout* t) {
    int64_t bo[13];
    block_param_offsets(c.W, bo);
    int64_t o = 0;
    t->tok = o; o += (int64_t)c.vocab * c.W;
    t->pos = o; o += (int64_t)c.L_ctx * c.W;
    t->blocks = o; t->block_size = bo[12]; o += bo[12] * c.layers;
    t->lnf_g = o; o += c.W;
    t->lnf_b = o; o += c.W;
    t->text_proj = o; o += (int64_t)c.W * c.D;
    t->n_params = o;
    for (int i = 0; i < 13; ++i) t->block_off[i] = bo[i];
    t->bf16_block_size = block_bf16_size(c.W);
    t->bf16_text_proj = t->bf16_block_size * c.layers;
    t->bf16_text_proj_t = t->bf16_text_proj + (int64_t)c.W * c.D;
    t->n_bf16 = t->bf16_text_proj_t + (int64_t)c.W * c.D;
}

// activation arena: [eot][cu][row_b][row_l][eot_row][per-layer block acts][x_final][e][mean_f][rstd_f][ln_e]
// Packed mode (c.T > 0): rows after each caption's EOT token are dead under the causal mask (they reach neither
// the pooled feature nor any gradient), so only the T live rows exist; cu/row_b/row_l/eot_row describe them.
struct TextActs {
    int32_t* eot;
    int32_t *cu, *row_b, *row_l, *eot_row;
    char* blocks;
    size_t block_bytes;
    float* x_final;
    float* e;
    float *mean_f, *rstd_f;
    bf16_t* ln_e;
    float* pool_xin;           // [B, W] the last block's residual stream at the pooled rows (pooled last block)
    bf16_t* pool_attn;         // [B, W] its attention output at the pooled rows
};

// Pooled last block (TextCfg.pool; SPN_POOL_LAST=0 switches it off whatever the caller asks for).  CLIP pools ONE row per
// caption - x[arange, argmax(ids)] behind ln_final (clip/model.py:352-356) - so of the LAST block only that row's output is
// ever read, and everything behind its attention is row-wise: the out-projection, ln_2 and the MLP (9 of the block's 12 W^2
// multiply-adds per token, forward and backward, and their three weight gradients) run on the B pooled rows instead of the
// B x L (or packed T) rows; the backward scatters the two gradients that re-enter the all-rows part (residual stream,
// attention output) and continues there.  Same kernels on the same rows: the features and gradients are those of the
// all-rows computation up to the summation order of the three weight gradients (B rows instead of T, of which T - B were zero).
static bool text_pooled(const TextCfg& c) {
    static const bool off = [] {
        const char* e = spn_env("SPN_POOL_LAST");
        return e && e[0] == '0';
    }();
    return c.pool != 0 && !off;
}

size_t text_act_bytes(const TextCfg& c) {
    const BlockCfg bc = text_block_cfg(c);
    const size_t T = (size_t)bc.rows();
    size_t b = 2 * align256((size_t)c.B * 4) + align256((size_t)(c.B + 1) * 4) + 2 * align256(T * 4);
    b += block_act_bytes(bc) * c.layers;
    b += align256(T * c.W * 4);
    b += align256((size_t)c.B * c.W * 4);
    b += 2 * align256((size_t)c.B * 4);
    b += align256((size_t)c.B * c.W * 2);
    b += align256((size_t)c.B * c.W * 4) + align256((size_t)c.B * c.W * 2);      // pool_xin, pool_attn
    return b;
}

static TextActs text_acts_at(char* base, const TextCfg& c) {
    const BlockCfg bc = text_block_cfg(c);
    const size_t T = (size_t)bc.rows();
    TextActs A;
    char* p = base;
    auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
    A.eot = (int32_t*)take((size_t)c.B * 4);
    A.cu = (int32_t*)take((size_t)(c.B + 1) * 4);
    A.row_b = (int32_t*)take(T * 4);
    A.row_l = (int32_t*)take(T * 4);
    A.eot_row = (int32_t*)take((size_t)c.B * 4);
    A.block_bytes = block_act_bytes(bc);
    A.blocks = p; p += A.block_bytes * c.layers;
    A.x_final = (float*)take(T * c.W * 4);
    A.e = (float*)take((size_t)c.B * c.W * 4);
    A.mean_f = (float*)take((size_t)c.B * 4);
    A.rstd_f = (float*)take((size_t)c.B * 4);
    A.ln_e = (bf16_t*)take((size_t)c.B * c.W * 2);
    A.pool_xin = (float*)take((size_t)c.B * c.W * 4);
    A.pool_attn = (bf16_t*)take((size_t)c.B * c.W * 2);
    return A;
}

static size_t tn_side_ws_bytes(const BlockCfg& c) {
    const int T = c.rows(), W = c.W;
    size_t m = 0;
    auto mx = [&](size_t v) { if (v > m) m = v; };
    mx(gemm_tn_workspace_bytes(T, W, 4 * W));
    mx(gemm_tn_workspace_bytes(T, 4 * W, W));
    mx(gemm_tn_workspace_bytes(T, W, W));
    mx(gemm_tn_workspace_bytes(T, 3 * W, W));
    return align256(m);
}

// SPN_BWD_OVERLAP=1 (opt-in) runs the weight-gradient GEMMs of the text tower's backward on a side stream.
// Measured on config 2: 19.0 -> 18.5 ms per step (+2.7 %), but the co-running kernels stretch each other
// (gemm_nt 83 -> 98 us, gemm_tn 90 -> 151 us), so per-kernel timings stop being meaningful; off by default.
// (Two processes sharing ONE GPU with it enabled crawl - the 2-rank single-GPU tests took 13 min instead of 4 s.)
// The side stream and its events are created once per process (one process per GPU).
static const BwdOverlap* bwd_overlap(bf16_t* dxb_alt, float* ws2, size_t ws2_bytes, BwdOverlap* out) {
    static const bool on = [] {
        const char* e = spn_env("SPN_BWD_OVERLAP");
        return e && e[0] == '1';
    }();
    if (!on) return nullptr;
    static BwdOverlap base;
    static bool ok = [] {
        if (hipStreamCreateWithFlags(&base.side, hipStreamNonBlocking) != hipSuccess) return false;
        for (int i = 0; i < 6; ++i)
            if (hipEventCreateWithFlags(&base.ev[i], hipEventDisableTiming) != hipSuccess) return false;
        return true;
    }();
    if (!ok) return nullptr;
    *out = base;
    out->dxb_alt = dxb_alt; out->ws2 = ws2; out->ws2_bytes = ws2_bytes;
    return out;
}

// per-layer buffers of the deferred backward: dpre [T,4W] | dqkv [T,3W] | dx_mid [T,W] | dx_in [T,W] (the gradient
// entering the block; the top block reads the head's buffer instead)
static size_t text_defer_layer_bytes(const BlockCfg& bc) {
    const size_t T = (size_t)bc.rows(), W = bc.W;
    return align256(T * 4 * W * 2) + align256(T * 3 * W * 2) + 2 * align256(T * W * 2) +
           align256(2 * layernorm_bwd_workspace_bytes((int)T, (int)W));
}

size_t text_ws_bytes(const TextCfg& c) {
    const BlockCfg bc = text_block_cfg(c);
    const size_t T = (size_t)bc.rows();
    size_t b = block_bwd_scratch_bytes(bc);
    b += align256(T * c.W * 4);              // dx
    b += 2 * align256(T * c.W * 2);          // dx_bf16 + its alternate (overlapped backward)
    b += align256(tn_side_ws_bytes(bc));     // split-K workspace of the side-stream weight-gradient GEMMs
    b += align256((size_t)c.B * c.D * 2);    // dfeats bf16
    b += align256((size_t)c.B * c.W * 4);    // dln_e
    b += align256((size_t)c.B * c.W * 4);    // de
    b += align256((size_t)c.B * 4 * c.W * 2) + 4 * align256((size_t)c.B * c.W * 2);   // pooled last block: dpre, de / dh / dx_mid / dattn (bf16, B rows)
    // per-layer dY operands of the deferred weight gradients (9 T W bf16 + LayerNorm partials per block: 3.3 GB for
    // ViT-L/14 text at B = 256, linear in the batch); absent when the deferred path is switched off (SPN_TN_GROUP=0:
    // spn_text_bwd_layer_deferred / spn_text_bwd_wgrad then return SPN_ERR_WORKSPACE, spn_text_bwd_layer needs none)
    b += text_defer_layer_bytes(bc) * (tn_group_on() ? c.layers : 0);
    size_t op = block_op_ws_bytes(bc);
    const size_t tp = align256(gemm_tn_workspace_bytes(c.B, c.W, c.D));
    if (tp > op) op = tp;
    const size_t lb = align256(layernorm_bwd_workspace_bytes(c.B, c.W));
    if (lb > op) op = lb;
    size_t eb = align256(embed_bwd_all_ws_bytes(c.B, c.L, c.W));         // text_bwd_tokens' embedding backward
    if (align256(embed_bwd_packed_ws_bytes(c.L, c.W)) > eb) eb = align256(embed_bwd_packed_ws_bytes(c.L, c.W));
    if (eb > op) op = eb;
    return b + op;
}

static int text_check(const TextCfg& c) {
    if (c.B <= 0 || c.L <= 0 || c.L > c.L_ctx || c.layers <= 0 || c.vocab <= 0) return SPN_ERR_ARG;
    if (c.T < 0 || (c.T > 0 && (c.T < c.B || (int64_t)c.T > (int64_t)c.B * c.L))) return SPN_ERR_ARG;
    if (c.T > 0 && c.L > 128) return SPN_ERR_SHAPE;   // packed rows need the whole-head attention kernels
    if (c.W % 64 || c.H * 64 != c.W || c.D % 64) return SPN_ERR_SHAPE;
    return SPN_OK;
}

int text_refresh_bf16(const TextCfg& c, const float* params, bf16_t* wb, hipStream_t st) {
    SPN_TRY(text_check(c));
    TextLayout t;
    text_layout(c, &t);
    SPN_TRY(blocks_refresh_bf16(params + t.blocks, t.block_size, wb, t.bf16_block_size, c.layers, c.W, st));
    SPN_TRY(cast_transpose_f32_bf16(params + t.text_proj, wb + t.bf16_text_proj, wb + t.bf16_text_proj_t, c.W, c.D, st));
    return SPN_OK;
}

// The last block with its row-wise half on the pooled rows only (text_pooled): A.e = the block output at the pooled rows.
static int text_last_block_fwd(const TextCfg& c, const BlockCfg& bc, const BlockParams& P, const BlockActs& a, const TextActs& A,
                               const float* x_prev, hipStream_t st) {
    const int B = c.B, W = c.W;
    SPN_TRY(block_fwd_attn(bc, P, a, st, x_prev));
    SPN_TRY(gather_pool_rows(a.x_in, a.attn, A.eot, c.T > 0 ? A.eot_row : nullptr, c.L, A.pool_xin, A.pool_attn, B, W, st));
    // from here on the first B rows of the block's own buffers (x_mid, h2, pre, u, mean2, rstd2) hold the pooled rows
    if (bc.fuse_resid) {
        GemmEpilogue e;
        e.bias = P.b_o; e.out_bf16 = a.h2; e.ldc = W;
        SPN_TRY(gemm_nt(A.pool_attn, P.w_o, B, W, W, W, W, GEMM_STORE, e, st));
        SPN_TRY(layernorm_fwd_add(A.pool_xin, a.h2, P.ln2_g, P.ln2_b, a.x_mid, a.h2, a.mean2, a.rstd2, B, W, bc.eps, st));
    } else {
        GemmEpilogue e;
        e.bias = P.b_o; e.resid = A.pool_xin; e.ldr = W; e.out_f32 = a.x_mid; e.ldc = W;
        SPN_TRY(gemm_nt(A.pool_attn, P.w_o, B, W, W, W, W, GEMM_RESID, e, st));
        SPN_TRY(layernorm_fwd(a.x_mid, P.ln2_g, P.ln2_b, a.h2, nullptr, a.mean2, a.rstd2, B, W, bc.eps, st));
    }
    {
        GemmEpilogue e;
        e.bias = P.b_fc; e.act = bc.act; e.aux_out = a.pre; e.aux_grad = aux_grad_on(); e.out_bf16 = a.u; e.ldc = 4 * W;
        SPN_TRY(gemm_nt(a.h2, P.w_fc, B, 4 * W, W, W, W, GEMM_STORE, e, st));
    }
    GemmEpilogue e;
    e.bias = P.b_proj; e.resid = a.x_mid; e.ldr = W; e.out_f32 = A.e; e.ldc = W;
    return gemm_nt(a.u, P.w_proj, B, W, 4 * W, 4 * W, 4 * W, GEMM_RESID, e, st);
}

int text_fwd(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, const int32_t* cu_seqlens,
             char* acts, float* feats, hipStream_t st) {
    SPN_TRY(text_check(c));
    if ((c.T > 0) != (cu_seqlens != nullptr)) return SPN_ERR_ARG;
    TextLayout t;
    text_layout(c, &t);
    BlockCfg bc = text_block_cfg(c);
    TextActs A = text_acts_at(acts, c);
    BlockActs first = block_acts_at(A.blocks, bc);
    if (c.T > 0) {
        // the prefix sums live in the arena from here on, so the backward phases need no extra argument
        bc.cu = A.cu;
        SPN_TRY(build_row_map(cu_seqlens, A.row_b, A.row_l, A.eot_row, c.B, st, cu_seqlens == A.cu ? nullptr : A.cu));
        SPN_TRY(embed_fwd_packed(ids, A.row_b, A.row_l, params + t.tok, params + t.pos, first.x_in, c.T, c.L, c.W, c.vocab,
                                 st));
    } else {
        SPN_TRY(eot_argmax(ids, A.eot, c.B, c.L, st));
        SPN_TRY(embed_fwd(ids, params + t.tok, params + t.pos, first.x_in, c.B, c.L, c.W, c.vocab, st));
    }
    bc.fuse_resid = fuse_resid_on();
    const bool pooled = text_pooled(c);
    const float* x_prev = nullptr;             // fuse_resid: the previous block's x_mid, its c_proj result waits in a.h1
    for (int l = 0; l < c.layers; ++l) {
        BlockActs a = block_acts_at(A.blocks + A.block_bytes * l, bc);
        const bool last = l + 1 == c.layers;
        BlockActs nxt = last ? a : block_acts_at(A.blocks + A.block_bytes * (l + 1), bc);
        a.x_out = last ? A.x_final : nxt.x_in;
        const BlockParams P = block_params_at(params + t.blocks + t.block_size * l, wb + t.bf16_block_size * l, c.W);
        if (last && pooled) {
            SPN_TRY(text_last_block_fwd(c, bc, P, a, A, x_prev, st));
            break;
        }
        // the last block's c_proj keeps its fp32 residual epilogue: x_final is read row-wise by the pooling below
        SPN_TRY(block_fwd(bc, P, a, st, x_prev, (bc.fuse_resid && !last) ? nxt.h1 : nullptr));
        x_prev = (bc.fuse_resid && !last) ? a.x_mid : nullptr;
    }
    // ln_final is per-row, so pooling the EOT row first is identical to clip/model.py:352-356
    if (pooled) {}                                    // A.e is the last block's output at the pooled rows already
    else if (c.T > 0) SPN_TRY(gather_rows_abs(A.x_final, A.eot_row, A.e, c.B, c.W, st));
    else SPN_TRY(gather_rows_f32(A.x_final, A.eot, A.e, c.B, c.L, c.W, st));
    SPN_TRY(layernorm_fwd(A.e, params + t.lnf_g, params + t.lnf_b, A.ln_e, nullptr, A.mean_f, A.rstd_f, c.B, c.W, 1e-5f,
                          st));
    GemmEpilogue e;
    e.out_f32 = feats; e.ldc = c.D;
    SPN_TRY(gemm_nt(A.ln_e, wb + t.bf16_text_proj_t, c.B, c.D, c.W, c.W, c.W, GEMM_STORE, e, st));
    return SPN_OK;
}

// Token output (tgcir/models.py:127-151, Backbone.extract_text_fea): ln_final over EVERY row - TG-CIR feeds all 77
// positions, padding included, to text_fc + TokenLearner - next to the pooled feature.  Dense layout only: the rows
// after the EOT token are live here.
int text_fwd_tokens(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts, float* feats,
                    float* tokens, bf16_t* tokens_bf16, float* tok_mean, float* tok_rstd, hipStream_t st) {
    if (c.T > 0 || c.pool) return SPN_ERR_ARG;      // ln_final of EVERY row: the last block cannot be pooled
    SPN_TRY(text_fwd(c, params, wb, ids, nullptr, acts, feats, st));
    TextLayout t;
    text_layout(c, &t);
    TextActs A = text_acts_at(acts, c);
    return layernorm_fwd(A.x_final, params + t.lnf_g, params + t.lnf_b, tokens_bf16, tokens, tok_mean, tok_rstd, c.B * c.L,
                         c.W, 1e-5f, st);
}

// backward workspace carve-up (identical in every phase, so dx survives between the calls)
struct TextBwdWs {
    char* scratch;
    float* dx;
    bf16_t* dxb;
    bf16_t* dxb2;
    float* ws2;
    size_t ws2_bytes;
    bf16_t* dfb;
    float *dln, *de;
    bf16_t *p_dpre, *p_deb, *p_dh, *p_dxm, *p_dattn;      // pooled last block, B rows each
    char* defer;               // layers x text_defer_layer_bytes
    size_t defer_stride;
    float* opws;
    size_t opws_bytes;
};

struct TextDeferBufs { bf16_t *dpre, *dqkv, *dx_mid, *dx_in; float* ln_partials; };
static TextDeferBufs text_defer_at(const TextBwdWs& w, const BlockCfg& bc, int l) {
    const size_t T = (size_t)bc.rows(), W = bc.W;
    char* p = w.defer + w.defer_stride * l;
    TextDeferBufs d;
    d.dpre = (bf16_t*)p; p += align256(T * 4 * W * 2);
    d.dqkv = (bf16_t*)p; p += align256(T * 3 * W * 2);
    d.dx_mid = (bf16_t*)p; p += align256(T * W * 2);
    d.dx_in = (bf16_t*)p; p += align256(T * W * 2);
    d.ln_partials = (float*)p;
    return d;
}

static int text_bwd_ws(const TextCfg& c, char* ws, size_t ws_bytes, TextBwdWs* w) {
    if (ws_bytes < text_ws_bytes(c)) return SPN_ERR_WORKSPACE;
    const BlockCfg bc = text_block_cfg(c);
    const size_t T = (size_t)bc.rows();
    char* p = ws;
    auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
    w->scratch = p; p += block_bwd_scratch_bytes(bc);
    w->dx = (float*)take(T * c.W * 4);
    w->dxb = (bf16_t*)take(T * c.W * 2);
    w->dxb2 = (bf16_t*)take(T * c.W * 2);
    w->ws2_bytes = tn_side_ws_bytes(bc);
    w->ws2 = (float*)take(w->ws2_bytes);
    w->dfb = (bf16_t*)take((size_t)c.B * c.D * 2);
    w->dln = (float*)take((size_t)c.B * c.W * 4);
    w->de = (float*)take((size_t)c.B * c.W * 4);
    w->p_dpre = (bf16_t*)take((size_t)c.B * 4 * c.W * 2);
    w->p_deb = (bf16_t*)take((size_t)c.B * c.W * 2);
    w->p_dh = (bf16_t*)take((size_t)c.B * c.W * 2);
    w->p_dxm = (bf16_t*)take((size_t)c.B * c.W * 2);
    w->p_dattn = (bf16_t*)take((size_t)c.B * c.W * 2);
    w->defer_stride = text_defer_layer_bytes(bc);
    w->defer = tn_group_on() ? p : nullptr;
    p += w->defer_stride * (tn_group_on() ? c.layers : 0);
    w->opws = (float*)p;
    w->opws_bytes = ws_bytes - (size_t)(p - ws);
    return SPN_OK;
}

// phase 1: text_projection, ln_final, scatter of the EOT-row gradient into dx
int text_bwd_head(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, const float* dfeats,
                  float* grads, char* ws, size_t ws_bytes, hipStream_t st) {
    SPN_TRY(text_check(c));
    TextBwdWs w;
    SPN_TRY(text_bwd_ws(c, ws, ws_bytes, &w));
    TextLayout t;
    text_layout(c, &t);
    TextActs A = text_acts_at(acts, c);
    SPN_TRY(cast_f32_bf16(dfeats, w.dfb, (size_t)c.B * c.D, st));
    SPN_TRY(gemm_tn(A.ln_e, w.dfb, c.B, c.W, c.D, c.W, c.D, grads + t.text_proj, c.D, 1.0f, 0, nullptr, w.opws, w.opws_bytes, st));
    {
        GemmEpilogue e;
        e.out_f32 = w.dln; e.ldc = c.W;
        SPN_TRY(gemm_nt(w.dfb, wb + t.bf16_text_proj, c.B, c.W, c.D, c.D, c.D, GEMM_STORE, e, st));
    }
    SPN_TRY(layernorm_bwd(nullptr, w.dln, A.e, params + t.lnf_g, A.mean_f, A.rstd_f, w.de, 0, nullptr, grads + t.lnf_g,
                          grads + t.lnf_b, 0, c.B, c.W, w.opws, w.opws_bytes, st));
    if (text_pooled(c)) return SPN_OK;          // the last block takes w.de as it is (text_last_block_bwd scatters behind its MLP)
    if (c.T > 0) SPN_TRY(scatter_rows_abs(w.de, A.row_b, A.eot_row, w.dx, w.dxb, c.T, c.W, st));
    else SPN_TRY(scatter_rows_f32(w.de, A.eot, w.dx, w.dxb, c.B, c.L, c.W, st));
    return SPN_OK;
}

// Backward of text_last_block_fwd.  In: w.de = gradient w.r.t. the pooled rows of the block output [B, W].  The row-wise half
// runs on the B pooled rows with its three weight gradients finished on the spot (reduction over B rows); then the residual and
// attention-output gradients are scattered to their rows (zeros elsewhere) and the attention half runs over all rows.
// `defer` != null: the qkv weight gradient and ln_1's parameter gradients are left to text_bwd_wgrad (dqkv and the row
// partials stay in the layer's deferred buffers); dx_out = the bf16 gradient leaving the block.
static int text_last_block_bwd(const TextCfg& c, const BlockCfg& bc, const BlockParams& P, const BlockActs& a, const BlockGrads& G,
                               const TextActs& A, const TextBwdWs& w, const TextDeferBufs* defer, bf16_t* dx_out, hipStream_t st) {
    const int B = c.B, W = c.W, T = bc.rows();
    const size_t Ts = (size_t)T;
    char* p = w.scratch;
    auto take = [&](size_t bytes) { char* r = p; p += align256(bytes); return r; };
    take(Ts * 4 * W * 2);                                     // dpre of the all-rows form: unused here
    bf16_t* dh = (bf16_t*)take(Ts * W * 2);
    bf16_t* dattn = (bf16_t*)take(Ts * W * 2);
    bf16_t* dqkv = (bf16_t*)take(Ts * 3 * W * 2);
    float* delta = (float*)take((size_t)c.B * c.H * c.L * 4);
    if (defer) dqkv = defer->dqkv;
    // ---- row-wise half on the pooled rows
    SPN_TRY(cast_f32_bf16(w.de, w.p_deb, (size_t)B * W, st));
    {
        GemmEpilogue e;
        e.aux_in = a.pre; e.aux_grad = aux_grad_on(); e.act = bc.act; e.out_bf16 = w.p_dpre; e.ldc = 4 * W;
        SPN_TRY(gemm_nt(w.p_deb, P.w_proj_t, B, 4 * W, W, W, W, GEMM_DACT, e, st));
    }
    SPN_TRY(gemm_tn(w.p_deb, a.u, B, W, 4 * W, W, 4 * W, G.w_proj, 4 * W, 1.0f, 0, G.b_proj, w.opws, w.opws_bytes, st));
    {
        GemmEpilogue e;
        e.out_bf16 = w.p_dh; e.ldc = W;
        SPN_TRY(gemm_nt(w.p_dpre, P.w_fc_t, B, W, 4 * W, 4 * W, 4 * W, GEMM_STORE, e, st));
    }
    SPN_TRY(gemm_tn(w.p_dpre, a.h2, B, 4 * W, W, 4 * W, W, G.w_fc, W, 1.0f, 0, G.b_fc, w.opws, w.opws_bytes, st));
    SPN_TRY(layernorm_bwd(w.p_dh, nullptr, a.x_mid, P.ln2_g, a.mean2, a.rstd2, w.de, 1, w.p_dxm, G.ln2_g, G.ln2_b, 0, B, W, w.opws,
                          w.opws_bytes, st));
    {
        GemmEpilogue e;
        e.out_bf16 = w.p_dattn; e.ldc = W;
        SPN_TRY(gemm_nt(w.p_dxm, P.w_o_t, B, W, W, W, W, GEMM_STORE, e, st));
    }
    SPN_TRY(gemm_tn(w.p_dxm, A.pool_attn, B, W, W, W, W, G.w_o, W, 1.0f, 0, G.b_o, w.opws, w.opws_bytes, st));
    // ---- back to all rows
    SPN_TRY(scatter_pool_rows(w.de, w.p_dattn, A.eot, c.T > 0 ? A.row_b : nullptr, A.eot_row, c.L, w.dx, dattn, T, W, st));
    {
        AttnBwdArgs g;
        AttnArgs& q = g.f;
        q.q = a.qkv; q.k = a.qkv + W; q.v = a.qkv + 2 * W;
        q.ldq = q.ldk = q.ldv = 3 * W;
        q.o = a.attn; q.ldo = W; q.lse = a.lse; q.key_bias = nullptr;
        q.B = bc.B; q.H = bc.H; q.Lq = bc.L; q.Lk = bc.L; q.causal = bc.causal; q.cu = bc.cu;
        q.scale = 0.125f;
        g.d_o = dattn; g.lddo = W;
        g.dq = dqkv; g.dk = dqkv + W; g.dv = dqkv + 2 * W;
        g.lddq = g.lddk = g.lddv = 3 * W;
        g.delta = delta;
        SPN_TRY(attention_bwd(g, st));
    }
    {
        GemmEpilogue e;
        e.out_bf16 = dh; e.ldc = W;
        SPN_TRY(gemm_nt(dqkv, P.w_qkv_t, T, W, 3 * W, 3 * W, 3 * W, GEMM_STORE, e, st));
    }
    const size_t lnp = layernorm_bwd_workspace_bytes(T, W);
    if (defer)
        return layernorm_bwd(dh, nullptr, a.x_in, P.ln1_g, a.mean1, a.rstd1, w.dx, 1, dx_out, G.ln1_g, G.ln1_b, 2, T, W,
                             (float*)((char*)defer->ln_partials + lnp), lnp, st);
    SPN_TRY(gemm_tn(dqkv, a.h1, T, 3 * W, W, 3 * W, W, G.w_qkv, W, 1.0f, 0, G.b_qkv, w.opws, w.opws_bytes, st));
    return layernorm_bwd(dh, nullptr, a.x_in, P.ln1_g, a.mean1, a.rstd1, w.dx, 1, dx_out, G.ln1_g, G.ln1_b, 0, T, W, w.opws,
                         w.opws_bytes, st);
}

// phase 2 (layers-1 .. 0): one residual block; its parameter gradients are final on return
int text_bwd_layer(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, float* grads, int l, char* ws,
                   size_t ws_bytes, hipStream_t st) {
    SPN_TRY(text_check(c));
    if (l < 0 || l >= c.layers) return SPN_ERR_ARG;
    TextBwdWs w;
    SPN_TRY(text_bwd_ws(c, ws, ws_bytes, &w));
    TextLayout t;
    text_layout(c, &t);
    BlockCfg bc = text_block_cfg(c);
    TextActs A = text_acts_at(acts, c);
    if (c.T > 0) bc.cu = A.cu;
    BlockActs a = block_acts_at(A.blocks + A.block_bytes * l, bc);
    const BlockParams P = block_params_at(params + t.blocks + t.block_size * l, wb + t.bf16_block_size * l, c.W);
    const BlockGrads G = block_grads_at(grads + t.blocks + t.block_size * l, c.W);
    if (l == c.layers - 1 && text_pooled(c)) return text_last_block_bwd(c, bc, P, a, G, A, w, nullptr, w.dxb, st);
    BwdOverlap ovs;
    const BwdOverlap* ov = bwd_overlap(w.dxb2, w.ws2, w.ws2_bytes, &ovs);
    return block_bwd(bc, P, a, G, w.dx, w.dxb, w.scratch, w.opws, w.opws_bytes, st, ov, w.dxb2);
}

// phase 2, deferred: the block's data path only; its four weight-gradient products are left to text_bwd_wgrad
// (returns the number of problems written)
static int text_defer_problems(const TextCfg& c, const TextBwdWs& w, const BlockCfg& bc, const TextActs& A, float* grads,
                               const TextLayout& t, int l, TnProblem* q) {
    const int W = c.W;
    const TextDeferBufs d = text_defer_at(w, bc, l);
    const BlockActs a = block_acts_at(A.blocks + A.block_bytes * l, bc);
    const BlockGrads G = block_grads_at(grads + t.blocks + t.block_size * l, c.W);
    const bf16_t* dx_in = (l == c.layers - 1) ? w.dxb : d.dx_in;
    if (l == c.layers - 1 && text_pooled(c)) {        // the pooled last block owes the qkv product only
        q[0] = TnProblem{d.dqkv, a.h1, G.w_qkv, G.b_qkv, 3 * W, W, 3 * W, W, W};
        return 1;
    }
    q[0] = TnProblem{dx_in, a.u, G.w_proj, G.b_proj, W, 4 * W, W, 4 * W, 4 * W};
    q[1] = TnProblem{d.dpre, a.h2, G.w_fc, G.b_fc, 4 * W, W, 4 * W, W, W};
    q[2] = TnProblem{d.dqkv, a.h1, G.w_qkv, G.b_qkv, 3 * W, W, 3 * W, W, W};
    q[3] = TnProblem{d.dx_mid, a.attn, G.w_o, G.b_o, W, W, W, W, W};
    return 4;
}

int text_bwd_layer_deferred(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, float* grads, int l,
                            char* ws, size_t ws_bytes, hipStream_t st) {
    SPN_TRY(text_check(c));
    if (l < 0 || l >= c.layers) return SPN_ERR_ARG;
    TextBwdWs w;
    SPN_TRY(text_bwd_ws(c, ws, ws_bytes, &w));
    if (!w.defer) return SPN_ERR_WORKSPACE;          // SPN_TN_GROUP=0: the workspace holds no deferred buffers
    TextLayout t;
    text_layout(c, &t);
    BlockCfg bc = text_block_cfg(c);
    TextActs A = text_acts_at(acts, c);
    if (c.T > 0) bc.cu = A.cu;
    BlockActs a = block_acts_at(A.blocks + A.block_bytes * l, bc);
    const BlockParams P = block_params_at(params + t.blocks + t.block_size * l, wb + t.bf16_block_size * l, c.W);
    const BlockGrads G = block_grads_at(grads + t.blocks + t.block_size * l, c.W);
    const TextDeferBufs d = text_defer_at(w, bc, l);
    if (l == c.layers - 1 && text_pooled(c))
        return text_last_block_bwd(c, bc, P, a, G, A, w, &d, l > 0 ? text_defer_at(w, bc, l - 1).dx_in : w.dxb2, st);
    TnProblem unused[4];
    BwdDefer df;
    df.dpre = d.dpre; df.dqkv = d.dqkv; df.dx_mid = d.dx_mid; df.ln_partials = d.ln_partials;
    df.dx_out = l > 0 ? text_defer_at(w, bc, l - 1).dx_in : w.dxb2;      // the gradient entering block l-1
    df.problems = unused;                                               // rebuilt by text_bwd_wgrad from the layout
    bf16_t* dx_in = (l == c.layers - 1) ? w.dxb : d.dx_in;
    return block_bwd(bc, P, a, G, w.dx, dx_in, w.scratch, w.opws, w.opws_bytes, st, nullptr, nullptr, &df);
}

int text_bwd_wgrad(const TextCfg& c, char* acts, float* grads, int l_begin, int l_end, char* ws, size_t ws_bytes,
                   hipStream_t st) {
    SPN_TRY(text_check(c));
    if (l_begin < 0 || l_end > c.layers || l_begin >= l_end || (l_end - l_begin) * 4 > TN_GROUP_MAX) return SPN_ERR_ARG;
    TextBwdWs w;
    SPN_TRY(text_bwd_ws(c, ws, ws_bytes, &w));
    if (!w.defer) return SPN_ERR_WORKSPACE;
    TextLayout t;
    text_layout(c, &t);
    BlockCfg bc = text_block_cfg(c);
    TextActs A = text_acts_at(acts, c);
    TnProblem q[TN_GROUP_MAX];
    int n = 0;
    for (int l = l_end - 1; l >= l_begin; --l) n += text_defer_problems(c, w, bc, A, grads, t, l, q + n);
    SPN_TRY(gemm_tn_grouped(q, n, bc.rows(), w.opws, w.opws_bytes, st));
    // the LayerNorm parameter gradients of the same blocks: one batched fold of their row partials ([dgamma | dbeta] are
    // adjacent in the flat gradient layout)
    const size_t lnp = layernorm_bwd_workspace_bytes(bc.rows(), c.W);
    FoldBatch fb{};
    fb.n = layernorm_bwd_partial_rows(bc.rows());
    fb.stride = (size_t)2 * c.W;
    fb.C = (size_t)2 * c.W;
    for (int l = l_end - 1; l >= l_begin; --l) {
        const TextDeferBufs d = text_defer_at(w, bc, l);
        const BlockGrads G = block_grads_at(grads + t.blocks + t.block_size * l, c.W);
        if (G.ln2_b != G.ln2_g + c.W || G.ln1_b != G.ln1_g + c.W) return SPN_ERR_ARG;
        if (!(l == c.layers - 1 && text_pooled(c))) {          // the pooled last block finished ln_2's gradients itself
            fb.ws[fb.items] = d.ln_partials; fb.out[fb.items++] = G.ln2_g;
        }
        fb.ws[fb.items] = (const float*)((const char*)d.ln_partials + lnp); fb.out[fb.items++] = G.ln1_g;
        if (fb.items + 2 > FOLD_BATCH_MAX || l == l_begin) {
            SPN_TRY(fold_rows_batched(fb, st));
            fb.items = 0;
        }
    }
    return SPN_OK;
}

// phase 3: token / positional embedding gradients
static int text_bwd_tail_impl(const TextCfg& c, const int32_t* ids, char* acts, float* grads, char* ws, size_t ws_bytes,
                              bool all_rows, hipStream_t st) {
    SPN_TRY(text_check(c));
    TextBwdWs w;
    SPN_TRY(text_bwd_ws(c, ws, ws_bytes, &w));
    TextLayout t;
    text_layout(c, &t);
    TextActs A = text_acts_at(acts, c);
    SPN_TRY(zero_fill_f32(grads + t.tok, (size_t)c.vocab * c.W, st));
    if (c.T > 0)
        SPN_TRY(embed_bwd_packed(ids, A.row_b, A.row_l, A.cu, w.dx, grads + t.tok, grads + t.pos, c.T, c.B, c.L, c.W, c.vocab,
                                 w.opws, w.opws_bytes, st));
    else if (all_rows)   // id 0 = CLIP's padding (clip/clip.py:236: zeros): the hot row
        SPN_TRY(embed_bwd_all(ids, w.dx, grads + t.tok, grads + t.pos, c.B, c.L, c.W, c.vocab, 0, w.opws, w.opws_bytes, st));
    else SPN_TRY(embed_bwd(ids, A.eot, w.dx, grads + t.tok, grads + t.pos, c.B, c.L, c.W, c.vocab, st));
    if (c.L < c.L_ctx) {
        SPN_TRY(zero_fill_f32(grads + t.pos + (size_t)c.L * c.W, (size_t)(c.L_ctx - c.L) * c.W, st));
    }
    return SPN_OK;
}

int text_bwd_tail(const TextCfg& c, const int32_t* ids, char* acts, float* grads, char* ws, size_t ws_bytes,
                  hipStream_t st) {
    return text_bwd_tail_impl(c, ids, acts, grads, ws, ws_bytes, false, st);
}

// backward of text_fwd_tokens: dfeats [B, D] (pooled feature) and dtokens [B*L, W] (ln_final output of every row)
// head of the token variant: text_bwd_head (dx = scatter of the EOT-row gradient) + ln_final backward of every row.  With
// text_bwd_layer[_deferred] / text_bwd_wgrad and text_bwd_tail_tokens this is text_bwd_tokens in phases (TG-CIR under data
// parallelism: a layer group's range goes to the all-reduce while the layers below still run).
int text_bwd_tokens_head(const TextCfg& c, const float* params, const bf16_t* wb, char* acts, const float* dfeats,
                         const float* dtokens, const float* tok_mean, const float* tok_rstd, float* grads, char* ws,
                         size_t ws_bytes, hipStream_t st) {
    if (c.T > 0 || c.pool) return SPN_ERR_ARG;
    SPN_TRY(text_bwd_head(c, params, wb, acts, dfeats, grads, ws, ws_bytes, st));
    TextBwdWs w;
    SPN_TRY(text_bwd_ws(c, ws, ws_bytes, &w));
    TextLayout t;
    text_layout(c, &t);
    TextActs A = text_acts_at(acts, c);
    return layernorm_bwd(nullptr, dtokens, A.x_final, params + t.lnf_g, tok_mean, tok_rstd, w.dx, 1, w.dxb, grads + t.lnf_g,
                         grads + t.lnf_b, 1, c.B * c.L, c.W, w.opws, w.opws_bytes, st);
}

int text_bwd_tail_tokens(const TextCfg& c, const int32_t* ids, char* acts, float* grads, char* ws, size_t ws_bytes,
                         hipStream_t st) {
    if (c.T > 0) return SPN_ERR_ARG;
    return text_bwd_tail_impl(c, ids, acts, grads, ws, ws_bytes, true, st);
}

int text_bwd_tokens(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
                    const float* dfeats, const float* dtokens, const float* tok_mean, const float* tok_rstd, float* grads,
                    char* ws, size_t ws_bytes, hipStream_t st) {
    if (c.T > 0) return SPN_ERR_ARG;
    // dx = scatter of the EOT-row gradient + ln_final backward of every row, accumulated into dx (and its bf16 mirror)
    SPN_TRY(text_bwd_tokens_head(c, params, wb, acts, dfeats, dtokens, tok_mean, tok_rstd, grads, ws, ws_bytes, st));
    if (!tn_group_on()) {
        for (int l = c.layers - 1; l >= 0; --l) SPN_TRY(text_bwd_layer(c, params, wb, acts, grads, l, ws, ws_bytes, st));
    } else {       // as text_bwd: data path of every block first, then the weight gradients in grouped launches
        for (int l = c.layers - 1; l >= 0; --l)
            SPN_TRY(text_bwd_layer_deferred(c, params, wb, acts, grads, l, ws, ws_bytes, st));
        constexpr int PER = TN_GROUP_MAX / 4;
        for (int e = c.layers; e > 0; e -= PER)
            SPN_TRY(text_bwd_wgrad(c, acts, grads, e > PER ? e - PER : 0, e, ws, ws_bytes, st));
    }
    return text_bwd_tail_impl(c, ids, acts, grads, ws, ws_bytes, true, st);
}

int text_bwd(const TextCfg& c, const float* params, const bf16_t* wb, const int32_t* ids, char* acts,
             const float* dfeats, float* grads, char* ws, size_t ws_bytes, hipStream_t st) {
    SPN_TRY(text_bwd_head(c, params, wb, acts, dfeats, grads, ws, ws_bytes, st));
    if (!tn_group_on()) {
        for (int l = c.layers - 1; l >= 0; --l) SPN_TRY(text_bwd_layer(c, params, wb, acts, grads, l, ws, ws_bytes, st));
        return text_bwd_tail(c, ids, acts, grads, ws, ws_bytes, st);
    }
    // data path of every block first, then the weight gradients of up to 12 blocks per grouped launch
    for (int l = c.layers - 1; l >= 0; --l)
        SPN_TRY(text_bwd_layer_deferred(c, params, wb, acts, grads, l, ws, ws_bytes, st));
    constexpr int PER = TN_GROUP_MAX / 4;
    for (int e = c.layers; e > 0; e -= PER)
        SPN_TRY(text_bwd_wgrad(c, acts, grads, e > PER ? e - PER : 0, e, ws, ws_bytes, st));
    return text_bwd_tail(c, ids, acts, grads, ws, ws_bytes, st);
}

}  // namespace spn
