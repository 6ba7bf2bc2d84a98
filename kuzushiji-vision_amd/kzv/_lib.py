"""ctypes binding of libkzv.so (include/kzv.h).  There is NO fallback: if the HIP library is missing
the product path raises -- the oracle under oracle/ is never used from here."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KZV_LIB") or os.path.join(_HERE, "libkzv.so")      # KZV_LIB: dev knob (A/B of two builds on one box)


class KzvError(RuntimeError):
    pass


class kzv_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "image_h", "image_w", "patch_h", "patch_w", "channels",
        "enc_hidden", "enc_layers", "enc_heads", "enc_ffn",
        "dec_hidden", "dec_layers", "dec_heads", "dec_ffn",
        "vocab", "max_pos", "type_vocab", "pad_id")] + [(n, C.c_float) for n in (
        "enc_hidden_dropout", "enc_attn_dropout", "dec_hidden_dropout", "dec_attn_dropout", "ln_eps")]


class kzv_opt_step(C.Structure):
    _fields_ = [("lr_t", C.c_float), ("ckp1", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("weight_decay", C.c_float), ("bias_correction2", C.c_float),
                ("adaptive", C.c_int32), ("max_grad_norm", C.c_float), ("grad_scale", C.c_float), ("one_minus_beta2", C.c_float)]


class kzv_gemm_nt_args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("bias", C.c_void_p),
                ("resid", C.c_void_p), ("ldr", C.c_int64), ("aux", C.c_void_p), ("ldaux", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("n_valid", C.c_int32),
                ("drop_p", C.c_float), ("drop_key", C.c_uint32)]


class kzv_gemm_nt_fp8_args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64),
                ("a_scale", C.c_void_p), ("b_scale", C.c_void_p),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("bias", C.c_void_p),
                ("resid", C.c_void_p), ("ldr", C.c_int64), ("aux", C.c_void_p), ("ldaux", C.c_int64),
                ("c8", C.c_void_p), ("ldc8", C.c_int64), ("c8_qscale", C.c_void_p), ("c8_amax", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("n_valid", C.c_int32),
                ("drop_p", C.c_float), ("drop_key", C.c_uint32), ("c8_rowq", C.c_void_p)]


class kzv_gemm_rows_ln_args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("bias", C.c_void_p), ("aux", C.c_void_p), ("ldaux", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("n_valid", C.c_int32),
                ("ln_a", C.c_void_p), ("ln_a_gamma", C.c_void_p), ("ln_a_beta", C.c_void_p),
                ("ln_r", C.c_void_p), ("ln_r_gamma", C.c_void_p), ("ln_r_beta", C.c_void_p), ("eps", C.c_float)]


class kzv_beam_state(C.Structure):
    _fields_ = [("batch", C.c_int32), ("num_beams", C.c_int32), ("max_len", C.c_int32), ("vocab", C.c_int32), ("eos_id", C.c_int32),
                ("run_seq_in", C.c_void_p), ("run_seq_out", C.c_void_p), ("fin_seq_in", C.c_void_p), ("fin_seq_out", C.c_void_p),
                ("run_scores", C.c_void_p), ("fin_scores", C.c_void_p), ("fin_done", C.c_void_p), ("fin_len", C.c_void_p),
                ("unsatisfied", C.c_void_p)]


class kzv_gemm_tn_args(C.Structure):
    _fields_ = [("P", C.c_void_p), ("ldp", C.c_int64), ("Q", C.c_void_p), ("ldq", C.c_int64),
                ("OUT", C.c_void_p), ("ldo", C.c_int64),
                ("Mtok", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("n_store", C.c_int32),
                ("dbias", C.c_void_p)]


class kzv_attn_args(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p), ("LSE", C.c_void_p),
                ("dO", C.c_void_p), ("dQ", C.c_void_p), ("dK", C.c_void_p), ("dV", C.c_void_p),
                ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
                ("ids", C.c_void_p), ("ld_ids", C.c_int64), ("pad_id", C.c_int32),
                ("B", C.c_int32), ("heads", C.c_int32), ("Sq", C.c_int32), ("Sk", C.c_int32), ("mode", C.c_int32),
                ("drop_p", C.c_float), ("drop_key", C.c_uint32), ("head_dim", C.c_int32)]


EPI_BF16, EPI_F32, EPI_GELU, EPI_RESID, EPI_DGELU = range(5)

# every symbol include/kzv.h declares: (restype, argtypes)
_P = C.c_void_p
class kzv_line_desc(C.Structure):
    _fields_ = [("src_off", C.c_int64), ("tmp_off", C.c_int64), ("hb_off", C.c_int64), ("hk_off", C.c_int64),
                ("vb_off", C.c_int64), ("vk_off", C.c_int64), ("in_h", C.c_int32), ("in_w", C.c_int32), ("new_h", C.c_int32),
                ("new_w", C.c_int32), ("paste_x", C.c_int32), ("paste_y", C.c_int32), ("hk_size", C.c_int32), ("vk_size", C.c_int32)]


SYMBOLS = {
    "kzv_last_error": (C.c_char_p, []),
    "kzv_version": (C.c_int, []),
    "kzv_model_create": (C.c_int, [C.POINTER(kzv_config), C.POINTER(_P)]),
    "kzv_model_destroy": (C.c_int, [_P]),
    "kzv_param_count": (C.c_int, [_P]),
    "kzv_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kzv_param_total": (C.c_int64, [_P]),
    "kzv_workspace_bytes": (C.c_int64, [_P, C.c_int, C.c_int]),
    "kzv_model_bind": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_int]),
    "kzv_model_sync_weights": (C.c_int, [_P, _P]),
    "kzv_forward_loss": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_uint64, _P]),
    "kzv_set_image_width": (C.c_int, [_P, C.c_int]),
    "kzv_check_positions": (C.c_int, [_P, _P]),
    "kzv_encode_images": (C.c_int, [_P, _P, C.c_int, _P]),
    "kzv_set_active_length": (C.c_int, [_P, C.c_int]),
    "kzv_decode_logits": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "kzv_zero_grads": (C.c_int, [_P, _P]),
    "kzv_backward_segments": (C.c_int, [_P]),
    "kzv_backward_segment": (C.c_int, [_P, C.c_int, _P]),
    "kzv_backward_segment_range": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "kzv_backward": (C.c_int, [_P, _P]),
    "kzv_grad_sqnorm": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "kzv_clip_and_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.POINTER(kzv_opt_step), _P]),
    "kzv_clip_and_step_ema": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, C.POINTER(kzv_opt_step), _P, C.c_float, _P]),
    "kzv_lerp_params": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "kzv_gemm_nt": (C.c_int, [C.POINTER(kzv_gemm_nt_args), C.c_int, _P]),
    "kzv_set_rows_max_m": (C.c_int, [C.c_int]),
    "kzv_set_nt_schedule": (C.c_int, [C.c_int]),
    "kzv_set_nt_half_epilogues": (C.c_int, [C.c_int]),
    "kzv_set_nt_half_stagger": (C.c_int, [C.c_int]),
    "kzv_set_tn_schedule": (C.c_int, [C.c_int]),
    "kzv_gemm_rows_ln": (C.c_int, [C.POINTER(kzv_gemm_rows_ln_args), C.c_int, _P]),
    "kzv_gemm_nt_fp8": (C.c_int, [C.POINTER(kzv_gemm_nt_fp8_args), C.c_int, _P]),
    "kzv_quant_rows_fp8": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P]),
    "kzv_layernorm_fwd_fp8": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "kzv_set_fp8": (C.c_int, [_P, C.c_int]),
    "kzv_get_fp8": (C.c_int, [_P]),
    "kzv_fp8_act_scales": (C.c_int, [_P, _P, _P]),
    "kzv_gemm_tn": (C.c_int, [C.POINTER(kzv_gemm_tn_args), _P]),
    "kzv_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "kzv_layernorm_bwd": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "kzv_attn_fwd": (C.c_int, [C.POINTER(kzv_attn_args), _P]),
    "kzv_attn_bwd": (C.c_int, [C.POINTER(kzv_attn_args), _P]),
    "kzv_drop_key": (C.c_uint32, [C.c_uint64, C.c_uint32]),
    "kzv_debug_dropout_mask": (C.c_int, [C.c_uint32, C.c_float, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "kzv_debug_attn_dropout_mask": (C.c_int, [C.c_uint32, C.c_float, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    # ---- ocr_lightning/model.py path (csrc/ocr.hip)
    "kzv_gemm_dgrad_wgrad": (C.c_int, [C.POINTER(kzv_gemm_nt_args), C.c_int, C.POINTER(kzv_gemm_tn_args), _P]),
    "kzv_set_pair": (C.c_int, [C.c_int]),
    "kzv_gemm_nt_f32": (C.c_int, [C.POINTER(kzv_gemm_nt_args), C.c_int, _P]),
    "kzv_gemm_tn_f32": (C.c_int, [C.POINTER(kzv_gemm_tn_args), _P]),
    "kzv_ocr_set_precision": (C.c_int, [C.c_int]),
    "kzv_ocr_nchw_to_nhwc": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "kzv_ocr_im2col": (C.c_int, [_P, _P] + [C.c_int] * 9 + [_P]),
    "kzv_ocr_col2im": (C.c_int, [_P, _P] + [C.c_int] * 10 + [_P]),
    "kzv_ocr_conv_weight": (C.c_int, [_P, _P, _P] + [C.c_int] * 5 + [_P]),
    "kzv_ocr_conv_wgrad_unpack": (C.c_int, [_P, _P] + [C.c_int] * 5 + [_P]),
    "kzv_ocr_conv_weight_multi": (C.c_int, [C.c_int, _P, _P, _P, _P, _P]),
    "kzv_ocr_conv_wgrad_unpack_multi": (C.c_int, [C.c_int, _P, _P, _P, _P]),
    "kzv_ocr_bn_fwd": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P]),
    "kzv_ocr_bn_bwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "kzv_ocr_bn_scratch_floats": (C.c_int64, [C.c_int64, C.c_int]),
    "kzv_ocr_maxpool_fwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "kzv_ocr_maxpool_bwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "kzv_ocr_avgpool_fwd": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "kzv_ocr_avgpool_bwd": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "kzv_ocr_lstm_cell_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "kzv_ocr_lstm_cell_bwd": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int, C.c_int, _P]),
    "kzv_ocr_log_softmax": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "kzv_ocr_ctc": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "kzv_ocr_smooth_l1_boxes": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, _P, _P, _P]),
    "kzv_ocr_adam": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _P]),
    "kzv_ocr_adam_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "kzv_ocr_cast_bf16": (C.c_int, [_P, _P, C.c_int64, _P]),
    "kzv_ocr_cast_transpose": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "kzv_prof_enable": (C.c_int, [C.c_int, C.c_int]),
    "kzv_prof_select": (C.c_int, [C.c_uint]),
    "kzv_prof_sample": (C.c_int, [C.c_int]),
    "kzv_prof_seen": (C.c_int64, [C.c_int]),
    "kzv_set_cu_reserve": (C.c_int, [C.c_int]),
    "kzv_decode_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "kzv_decode_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kzv_set_decode_one_launch": (C.c_int, [C.c_int]),
    "kzv_set_dec_chain": (C.c_int, [C.c_int]),
    "kzv_set_head_ce": (C.c_int, [C.c_int]),
    "kzv_decode_step_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "kzv_decode_reorder": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "kzv_decode_prep": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "kzv_greedy_update": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    "kzv_beam_update": (C.c_int, [C.POINTER(kzv_beam_state), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kzv_beam_topk": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kzv_lanczos_coeffs": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "kzv_preprocess_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "kzv_prof_collect": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libkzv.so and bind every declared symbol.  Raises KzvError loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KzvError(f"{LIB_PATH} not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch first: it ships its own libamdhip64; loaded after libkzv.so the process would hold two HIP runtimes (torch's
    # tensors in one, libkzv's launches in the other -> "no ROCm-capable device is detected" at the first launch)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().kzv_last_error().decode("utf-8", "replace")
        raise KzvError(f"{what or 'libkzv'} failed ({rc}): {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
