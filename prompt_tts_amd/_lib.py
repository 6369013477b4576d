"""ctypes binding of the C-ABI library (include/prompt_tts_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is absent,
import fails loudly.  Nothing here touches ``oracle/``.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the CDLL below: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7, and the
#               library here resolves the same soname.  Loaded after torch it shares torch's HIP runtime (streams and device
#               pointers are then the same objects); loaded BEFORE torch the process ends up with two HIP runtimes and every
#               launch here fails with "no ROCm-capable device is detected".

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_TTS_LIB: another build of the SAME library (tests load the host-sanitizer build, csrc `make asan`, this way)
LIB_PATH = os.environ.get("PT_TTS_LIB") or os.path.join(_HERE, "libprompt_tts_hip.so")

PT_F32, PT_BF16, PT_BF16X2 = 0, 1, 2
PT_FP8_E4M3, PT_FP8_E5M2 = 0, 1
PT_FP8_STATE_FLOATS = 258
PT_V_PLAIN, PT_V_CONCAT, PT_V_CONV, PT_V_WFLIP = 0, 1, 2, 3
PT_MAP_S1, PT_MAP_S2, PT_MAP_UP2, PT_MAP_S2_DGRAD, PT_MAP_CAUSAL_REFLECT, PT_MAP_BACK, PT_MAP_STRIDED_REFLECT = 0, 1, 2, 3, 4, 5, 6
PT_OUT_T, PT_OUT_F32, PT_OUT_F32_ATOMIC = 0, 1, 2


class pt_operand(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ld", C.c_int64), ("p2", C.c_void_p), ("ld2", C.c_int64),
                ("c_split", C.c_int64), ("kind", C.c_int32), ("trans", C.c_int32), ("taps", C.c_int32),
                ("cin", C.c_int32), ("rowmap", C.c_int32), ("stride", C.c_int32),
                ("n_out", C.c_int64), ("n_in", C.c_int64)]


class pt_gemm_desc(C.Structure):
    _fields_ = [("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64), ("A", pt_operand), ("B", pt_operand),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("out_kind", C.c_int32), ("split_k", C.c_int32),
                ("bias", C.c_void_p), ("row_bias", C.c_void_p), ("row_bias_rows", C.c_int64), ("row_bias_ld", C.c_int64),
                ("residual", C.c_void_p), ("ldr", C.c_int64), ("residual2", C.c_void_p), ("ldr2", C.c_int64),
                ("conv_wgrad_cin", C.c_int32),
                ("conv_wgrad_cin_store", C.c_int32), ("alpha", C.c_float), ("act", C.c_int32), ("act2", C.c_int32),
                ("C2", C.c_void_p), ("ldc2", C.c_int64),
                ("arow_sum", C.c_void_p), ("arow_n", C.c_int64), ("arow_stride", C.c_int64), ("arow_rep", C.c_int32),
                ("f32_x3", C.c_int32), ("geglu_rows", C.c_int64), ("x2_block", C.c_int64)]


class pt_attn_desc(C.Structure):
    _fields_ = [("B", C.c_int64), ("H", C.c_int64), ("Nq", C.c_int64), ("Nk", C.c_int64), ("D", C.c_int64),
                ("q", C.c_void_p), ("ldq", C.c_int64), ("k", C.c_void_p), ("ldk", C.c_int64),
                ("v", C.c_void_p), ("ldv", C.c_int64), ("o", C.c_void_p), ("ldo", C.c_int64),
                ("lse", C.c_void_p), ("scale", C.c_float), ("causal", C.c_int32), ("kv_len", C.c_void_p),
                ("d_o", C.c_void_p), ("lddo", C.c_int64), ("delta", C.c_void_p),
                ("dq", C.c_void_p), ("lddq", C.c_int64), ("dk", C.c_void_p), ("lddk", C.c_int64),
                ("dv", C.c_void_p), ("lddv", C.c_int64)]


class pt_rowconv_desc(C.Structure):
    _fields_ = [("B", C.c_int64), ("n_rows", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64), ("cin", C.c_int32),
                ("taps", C.c_int32), ("rowmap", C.c_int32), ("elu_x", C.c_int32), ("x2", C.c_void_p), ("ldx2", C.c_int64),
                ("cin2", C.c_int32), ("elu_x2", C.c_int32), ("w", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p),
                ("N", C.c_int32), ("act", C.c_int32), ("y", C.c_void_p), ("ldy", C.c_int64), ("y_f32", C.c_int32),
                ("stride", C.c_int32), ("f32_x3", C.c_int32), ("_pad", C.c_int32)]


class pt_lstm2_desc(C.Structure):
    _fields_ = [("B", C.c_int64), ("T", C.c_int64), ("H", C.c_int64), ("x", C.c_void_p), ("xg0", C.c_void_p),
                ("whh0", C.c_void_p), ("wcat1", C.c_void_p), ("bias1", C.c_void_p), ("h0_seq", C.c_void_p),
                ("h1_seq", C.c_void_p), ("c0", C.c_void_p), ("c1", C.c_void_p), ("out_elu", C.c_void_p), ("status", C.c_void_p),
                ("exact_f32", C.c_int64), ("per_step", C.c_int64)]


class pt_encodec_tail_desc(C.Structure):
    _fields_ = [("B", C.c_int64), ("n", C.c_int64), ("cin", C.c_int32), ("cout", C.c_int32), ("r", C.c_int32), ("_pad", C.c_int32),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("wt", C.c_void_p), ("bt", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("wf", C.c_void_p), ("bf", C.c_void_p), ("wfin", C.c_void_p), ("bfin", C.c_void_p), ("wav", C.c_void_p)]


class pt_encodec_stage_desc(C.Structure):
    _fields_ = [("B", C.c_int64), ("n", C.c_int64), ("cin", C.c_int32), ("cout", C.c_int32), ("r", C.c_int32), ("_pad", C.c_int32),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("wt", C.c_void_p), ("bt", C.c_void_p), ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("wf", C.c_void_p), ("bf", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int64)]


class pt_decode_linear_desc(C.Structure):
    _fields_ = [("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("geglu", C.c_int32),
                ("w", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("y", C.c_void_p), ("ldy", C.c_int64), ("seg_cols", C.c_int64), ("y2", C.c_void_p), ("ld2", C.c_int64),
                ("y3", C.c_void_p), ("ld3", C.c_int64), ("t_dev", C.c_void_p), ("t_stride", C.c_int64)]


class pt_ar_embed_desc(C.Structure):
    _fields_ = [("prev", C.c_void_p), ("emb", C.c_void_p), ("pos", C.c_void_p), ("t_dev", C.c_void_p), ("out", C.c_void_p),
                ("B", C.c_int64), ("n_q", C.c_int64), ("bins", C.c_int64), ("dim", C.c_int64)]


class pt_sample_desc(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("ld", C.c_int64), ("uniforms", C.c_void_p), ("out", C.c_void_p), ("R", C.c_int64),
                ("V", C.c_int64), ("k", C.c_int64), ("temperature", C.c_float), ("dtype", C.c_int32)]


class pt_ar_advance_desc(C.Structure):
    _fields_ = [("idx", C.c_void_p), ("prev", C.c_void_p), ("codes", C.c_void_p), ("t_dev", C.c_void_p), ("kv_len", C.c_void_p),
                ("B", C.c_int64), ("n_q", C.c_int64), ("T", C.c_int64)]


class pt_row_select_desc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int64), ("t_dev", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_int64)]


class pt_op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("desc", C.c_void_p)]


PT_OP_DECODE_LINEAR, PT_OP_ATTN_FWD, PT_OP_AR_EMBED, PT_OP_SAMPLE_TOPK, PT_OP_AR_ADVANCE, PT_OP_ROW_SELECT = range(6)


class pt_transpose_seg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64), ("src_ld", C.c_int64),
                ("dst_ld", C.c_int64), ("tile_begin", C.c_int64)]


class pt_param_seg(C.Structure):
    _fields_ = [("offset", C.c_int64), ("numel", C.c_int64), ("shadow_offset", C.c_int64),
                ("layout", C.c_int32), ("cin", C.c_int32), ("cin_pad", C.c_int32), ("frozen", C.c_int32)]


class pt_fold_seg(C.Structure):
    _fields_ = [("rep_off", C.c_int64), ("rep_stride", C.c_int64), ("dst_off", C.c_int64), ("n", C.c_int32), ("pad", C.c_int32)]


_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> argtypes (all return int status unless noted); mirrors include/prompt_tts_hip.h one to one.
SIGNATURES = {
    "pt_gemm": [C.POINTER(pt_gemm_desc), _i32, _vp],
    "pt_wgrad_group": [C.POINTER(pt_gemm_desc), _i32, _vp, _i64, _i32, _vp],
    "pt_gemm_fp8": [C.POINTER(pt_gemm_desc), _i32, _vp, _vp, _vp],
    "pt_fp8_quantize": [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp],
    "pt_attn_fwd": [C.POINTER(pt_attn_desc), _i32, _vp],
    "pt_attn_bwd": [C.POINTER(pt_attn_desc), _i32, _vp],
    "pt_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _vp],
    "pt_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _i32, _vp],
    "pt_fold_replicas": [_vp, _vp, _vp, _i64, _i32, _i64, _vp],
    "pt_transpose_batch": [_vp, _i64, _i64, _i32, _vp],
    "pt_groupnorm_stats": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _i32, _vp],
    "pt_groupnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _f32, _i32, _i32, _vp],
    "pt_groupnorm_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _i32, _vp],
    "pt_groupnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _i64, _i64, _i64, _i64, _i64, _i32, _i32, _f32, _i32, _i32, _i64, _vp, _i64, _i32, _vp],
    "pt_geglu_fwd": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "pt_geglu_bwd": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "pt_silu_fwd": [_vp, _vp, _i64, _i32, _vp],
    "pt_silu_bwd": [_vp, _vp, _vp, _i64, _i32, _vp],
    "pt_add": [_vp, _vp, _vp, _i64, _i32, _vp],
    "pt_dropout": [_vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp],
    "pt_pairsum_rows": [_vp, _vp, _i64, _i64, _i32, _vp],
    "pt_colsum": [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i32, _i64, _i32, _vp],
    "pt_embedding_fwd": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_embedding_bwd": [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "pt_timestep_embedding": [_vp, _vp, _i64, _i64, _i32, _f32, _i32, _vp],
    "pt_add_noise": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_tokens_to_bct": [_vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_bct_to_tokens": [_vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_mse_loss": [_vp, _vp, _vp, _vp, _f32, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_sumsq": [_vp, _vp, _i64, _vp],
    "pt_adamw_step": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _i64,
                      _i32, _vp],
    "pt_adamw_step_range": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _i64,
                            _i32, _i32, _vp],
    "pt_import_params_range": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_pack_shadow": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "pt_rvq_decode": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp],
    "pt_ddpm_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
    "pt_rvq_search": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp],
    "pt_rowconv": [C.POINTER(pt_rowconv_desc), _i32, _vp],
    "pt_lstm2_forward": [C.POINTER(pt_lstm2_desc), _i32, _vp],
    "pt_encodec_tail": [C.POINTER(pt_encodec_tail_desc), _i32, _vp],
    "pt_encodec_stage": [C.POINTER(pt_encodec_stage_desc), _i32, _vp],
    "pt_encodec_res": [C.POINTER(pt_encodec_stage_desc), _i32, _vp],
    "pt_codes_from_continuous": [_vp, _vp, _i64, _i64, _vp],
    "pt_sample_topk": [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _vp],
    "pt_decode_linear": [C.POINTER(pt_decode_linear_desc), _vp],
    "pt_ar_embed": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "pt_ar_advance": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "pt_row_select": [_vp, _i64, _vp, _vp, _i64, _vp],
    "pt_run_ops": [C.POINTER(pt_op), _i64, _vp],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C prompt_tts_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.pt_abi_version.restype = C.c_int
    lib.pt_status_string.restype = C.c_char_p
    lib.pt_status_string.argtypes = [C.c_int]
    lib.pt_last_hip_error.restype = C.c_char_p
    lib.pt_struct_size.restype = C.c_int
    lib.pt_struct_size.argtypes = [C.c_int]
    lib.pt_wgrad_group_ws_floats.restype = C.c_int64
    lib.pt_wgrad_group_ws_floats.argtypes = [C.c_int]
    for i, st in enumerate((pt_operand, pt_gemm_desc, pt_attn_desc, pt_param_seg, pt_rowconv_desc, pt_lstm2_desc, pt_fold_seg, pt_encodec_tail_desc, pt_encodec_stage_desc, pt_transpose_seg, pt_decode_linear_desc)):
        if lib.pt_struct_size(i) != C.sizeof(st):
            raise ImportError(f"ctypes layout of {st.__name__} ({C.sizeof(st)} B) disagrees with the library ({lib.pt_struct_size(i)} B)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    return lib


lib = _load()


def check(status, what):
    if status != 0:
        hip = f" [{lib.pt_last_hip_error().decode()}]" if status == -3 else ""        # PT_ERR_LAUNCH: the HIP error behind it
        raise RuntimeError(f"{what}: {lib.pt_status_string(status).decode()} (status {status}){hip}")
