"""ctypes binding of libspn4cir_hip.so (the C-ABI declared in include/spn4cir_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is
absent, loading raises and every op fails loudly.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPN_LIB_PATH: load an experimental build of the same C-ABI instead (kernel A/B runs, tools/build_variant.sh)
LIB_PATH = os.environ.get("SPN_LIB_PATH") or os.path.join(_HERE, "libspn4cir_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "spn4cir_hip.h")

_lib = None

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class TextCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "L", "L_ctx", "W", "H", "layers", "D", "vocab", "T", "pool")]


class TextLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("tok", "pos", "blocks", "block_size", "lnf_g", "lnf_b", "text_proj",
                                         "n_params")] + [("block_off", C.c_int64 * 13)] + \
               [(n, C.c_int64) for n in ("bf16_block_size", "bf16_text_proj", "bf16_text_proj_t", "n_bf16")]


class VisionCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "res", "patch", "W", "H", "layers", "D", "kind")]


class VisionLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("conv1", "conv_b", "cls", "pos", "ln_pre_g", "ln_pre_b", "blocks",
                                         "block_size", "ln_post_g", "ln_post_b", "proj", "proj_b", "n_params")] + \
               [("block_off", C.c_int64 * 13)] + \
               [(n, C.c_int64) for n in ("bf16_conv1", "bf16_blocks", "bf16_block_size", "bf16_proj_t", "n_bf16", "kp",
                                         "seq", "bf16_proj")]


class FusionCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "L", "S", "W", "H", "layers", "I", "E", "Dp", "vocab", "max_pos", "T")]


class FusionLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("word", "pos", "emb_ln_g", "emb_ln_b", "layers", "layer_size", "proj_w",
                                         "proj_b", "n_params")] + [("layer_off", C.c_int64 * 21)] + \
               [(n, C.c_int64) for n in ("bf16_layer_size", "bf16_proj", "bf16_proj_t", "n_bf16")] + \
               [("bf16_off", C.c_int64 * 15)]


_SIGS = {
    "spn_abi_version": (i32, []),
    "spn_error_string": (C.c_char_p, [i32]),
    "spn_gemm_nt": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp]),
    "spn_gemm_nt_resid": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp]),
    "spn_gemm_nt_dact": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp]),
    "spn_gemm_tn": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, f32, i32, vp, vp, sz, vp]),
    "spn_gemm_tn_workspace_bytes": (sz, [i32, i32, i32]),
    "spn_gemm_tn_pair": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, i32, vp, sz, vp]),
    "spn_gemm_tn_pair_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "spn_gemm_tn_grouped": (i32, [vp, i32, i32, vp, sz, vp]),
    "spn_gemm_tn_grouped_workspace_bytes": (sz, [i32]),
    "spn_cast_f32_bf16": (i32, [vp, vp, sz, vp]),
    "spn_cast_bf16_f32": (i32, [vp, vp, sz, vp]),
    "spn_sum_ranks_bf16": (i32, [vp, i32, sz, vp, vp]),
    "spn_sum_ranks_f32": (i32, [vp, i32, sz, vp, vp]),
    "spn_jpeg_decode_batch": (i32, [vp, vp, i32, vp, i32, vp, vp, vp, sz, vp, vp, i32, i32, vp]),
    "spn_cast_transpose_f32_bf16": (i32, [vp, vp, vp, i32, i32, vp]),
    "spn_colsum_bf16": (i32, [vp, i32, i32, i32, vp, i32, vp, sz, vp]),
    "spn_colsum_workspace_bytes": (sz, [i32, i32]),
    "spn_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "spn_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "spn_layernorm_bwd_workspace_bytes": (sz, [i32, i32]),
    "spn_attention_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "spn_attention_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, vp,
                                i32, i32, i32, i32, i32, f32, vp]),
    "spn_embed_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spn_embed_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spn_combine_l2norm_fwd": (i32, [vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, vp]),
    "spn_combine_l2norm_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "spn_combine_l2norm_bwd_scaled": (i32, [vp, vp, vp, vp, vp, i32, i32, vp]),
    "spn_bank_stats_fwd": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, f32, vp, vp, sz, vp]),
    "spn_bank_loss_finalize": (i32, [vp, i32, i32, i64, f32, vp, vp, vp, vp]),
    "spn_bank_grad_q": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, f32, vp, f32, i64, f32, vp, vp, sz, vp]),
    "spn_bank_workspace_bytes": (sz, [i32, i32, i32]),
    "spn_bank_logits_bytes": (sz, [i32, i32]),
    "spn_bank_config": (i32, [i32]),
    "spn_config_dump": (i32, [C.c_char_p, i32]),
    "spn_gemm_config": (i32, [i32, i32]),
    "spn_bank_step_ok": (i32, [i32, i32, i32, i32]),
    "spn_negtype_workspace_bytes": (sz, [i32, i32]),
    "spn_negtype_head": (i32, [vp, vp, vp, i32, i32, f32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "spn_fusion_bwd_phase": (i32, [C.POINTER(FusionCfg), vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "spn_scale_cast_bf16": (i32, [vp, vp, i32, vp, i32, i32, i32, vp]),
    "spn_bank_step": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp]),
    "spn_bank_stats_fwd_save": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, sz, vp]),
    "spn_bank_grad_q_saved": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, f32, i64, f32, vp, vp, sz, vp]),
    "spn_bank_workspace_bytes_fp8": (sz, [i32, i32, i32]),
    "spn_bank_stats_fwd_tokmax": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, f32, vp, vp, sz, vp]),
    "spn_bank_grad_q_tokmax": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, f32, vp, f32, i64, f32, vp, vp, sz, vp]),
    "spn_adamw_step": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, f32, vp, vp]),
    "spn_adamw_step_scaled": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, vp, vp, vp]),
    "spn_adamw_tick": (i32, [vp, vp, vp]),
    "spn_adamw_step_dev": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, vp, vp, vp, vp]),
    "spn_grad_check_finite": (i32, [vp, sz, vp, vp]),
    "spn_cosine_scores_f64": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "spn_topk_from_scores": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "spn_text_layout": (i32, [C.POINTER(TextCfg), C.POINTER(TextLayout)]),
    "spn_text_act_bytes": (sz, [C.POINTER(TextCfg)]),
    "spn_text_ws_bytes": (sz, [C.POINTER(TextCfg)]),
    "spn_text_refresh_bf16": (i32, [C.POINTER(TextCfg), vp, vp, vp]),
    "spn_text_fwd": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp]),
    "spn_text_fwd_packed": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp, vp]),
    "spn_text_bwd": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "spn_text_fwd_tokens": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "spn_text_bwd_tokens": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "spn_tg_ws_bytes": (sz, [i32, i32]),
    "spn_tg_tokenlearn_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "spn_tg_tokenlearn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp]),
    "spn_tg_fuse_prep": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spn_tg_img_finish": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "spn_tg_gate_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "spn_tg_gate_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, vp]),
    "spn_tg_mod_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "spn_text_bwd_head": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp, sz, vp]),
    "spn_text_bwd_tokens_head": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "spn_text_bwd_tail_tokens": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, sz, vp]),
    "spn_text_bwd_layer": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, i32, vp, sz, vp]),
    "spn_text_bwd_tail": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, sz, vp]),
    "spn_text_bwd_layer_deferred": (i32, [C.POINTER(TextCfg), vp, vp, vp, vp, i32, vp, sz, vp]),
    "spn_text_bwd_wgrad": (i32, [C.POINTER(TextCfg), vp, vp, i32, i32, vp, sz, vp]),
    "spn_vision_layout": (i32, [C.POINTER(VisionCfg), C.POINTER(VisionLayout)]),
    "spn_vision_ws_bytes": (sz, [C.POINTER(VisionCfg)]),
    "spn_vision_refresh_bf16": (i32, [C.POINTER(VisionCfg), vp, vp, vp]),
    "spn_vision_fwd": (i32, [C.POINTER(VisionCfg), vp, vp, vp, vp, sz, vp, vp, vp]),
    "spn_vision_train_act_bytes": (sz, [C.POINTER(VisionCfg)]),
    "spn_vision_bwd_ws_bytes": (sz, [C.POINTER(VisionCfg)]),
    "spn_vision_fwd_train": (i32, [C.POINTER(VisionCfg), vp, vp, vp, vp, vp, vp]),
    "spn_vision_bwd": (i32, [C.POINTER(VisionCfg), vp, vp, vp, vp, vp, vp, sz, vp]),
    "spn_bank_quantize_fp8": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "spn_bank_dequant_fp8": (i32, [vp, vp, i32, i32, vp, vp]),
    "spn_bank_stats_fwd_fp8": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, sz, vp]),
    "spn_bank_grad_q_fp8": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, vp, f32, i64, f32, vp, vp, sz, vp]),
    "spn_preprocess_image": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "spn_text_exact_ws_bytes": (sz, [C.POINTER(TextCfg)]),
    "spn_text_fwd_exact": (i32, [C.POINTER(TextCfg), vp, vp, vp, sz, vp, vp]),
    "spn_vision_exact_ws_bytes": (sz, [C.POINTER(VisionCfg)]),
    "spn_vision_fwd_exact": (i32, [C.POINTER(VisionCfg), vp, vp, vp, sz, vp, vp]),
    "spn_gemm_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, f32, vp]),
    "spn_im2col3x3_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "spn_avgpool_nhwc_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "spn_attnpool_tokens_f32": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "spn_im2col3x3_nhwc_bf16": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "spn_im2col3x3_stem_bf16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "spn_relu_add_bf16": (i32, [vp, vp, C.c_size_t, vp]),
    "spn_avgpool_nhwc_bf16": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "spn_attnpool_attend_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "spn_inbatch_grad_t": (i32, [vp, vp, i32, vp, i32, i32, f32, f32, vp, vp]),
    "spn_fusion_layout": (i32, [C.POINTER(FusionCfg), C.POINTER(FusionLayout)]),
    "spn_fusion_packed_ok": (i32, [C.POINTER(FusionCfg)]),
    "spn_xattn_ok": (i32, [i32, i32, i32, i32, i32]),
    "spn_xattn_sp": (i32, [i32]),
    "spn_xattn_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "spn_xattn_bwd": (i32, [vp] * 17 + [i32, i32, i32, i32, i32, i32, f32, vp]),
    "spn_fusion_act_bytes": (sz, [C.POINTER(FusionCfg)]),
    "spn_fusion_ws_bytes": (sz, [C.POINTER(FusionCfg)]),
    "spn_fusion_refresh_bf16": (i32, [C.POINTER(FusionCfg), vp, vp, vp]),
    "spn_fusion_fwd": (i32, [C.POINTER(FusionCfg), vp, vp, vp, vp, vp, vp, vp, vp]),
    "spn_fusion_fwd_bank": (i32, [C.POINTER(FusionCfg), vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp]),
    "spn_gather_bank_rows_bf16": (i32, [vp, C.c_int64, vp, vp, i32, C.c_int64, vp]),
    "spn_tau_grad": (i32, [vp, vp, i32, vp, i32, i32, C.c_float, vp, vp, vp, vp]),
    "spn_fusion_bwd": (i32, [C.POINTER(FusionCfg), vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "spn_prof_enable": (i32, [i32]),
    "spn_prof_select": (i32, [C.c_uint, i32]),
    "spn_prof_disable": (i32, []),
    "spn_prof_reset": (i32, []),
    "spn_prof_collect": (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}


def header_symbols():
    """Every function name declared in include/spn4cir_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(spn_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load (once) and return the library; raises if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C spn4cir_amd/csrc`). spn4cir_amd has no CPU fallback.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(handle, name)          # AttributeError if the symbol is absent
        fn.restype, fn.argtypes = res, args
    if handle.spn_abi_version() != 1:
        raise RuntimeError("libspn4cir_hip.so ABI version mismatch")
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().spn_error_string(rc).decode()
        raise RuntimeError(f"spn4cir_hip {what} failed: {msg} (code {rc})")


_ENV_SNAPSHOT = None


def env(name, default=None):
    """SPN_* switch as the LIBRARY saw it when it was loaded (csrc/config.hip freezes the environment then and the kernels read
    only that snapshot).  The host side must route by the same value: a variable set after the load would otherwise make Python
    choose one call sequence (e.g. deferred / grouped weight gradients) while the library's own switch says the other."""
    global _ENV_SNAPSHOT
    if _ENV_SNAPSHOT is None:
        _ENV_SNAPSHOT = dict(config_dump().get("env", {}))
    return _ENV_SNAPSHOT.get(name, default)


def config_dump():
    """The library's run-time configuration as a dict: {"experiments_build": 0|1, "env": {every SPN_* variable present when
    the library was loaded}} - the kernels' A/B switches read that snapshot only (csrc/config.hip)."""
    import json
    L = lib()
    n = L.spn_config_dump(None, 0)
    buf = C.create_string_buffer(n)
    L.spn_config_dump(buf, n)
    return json.loads(buf.value.decode())
