"""ctypes binding of libesvit_hip.so (include/esvit_hip.h).

The product path has no fallback: if the shared library is missing or a symbol fails to
resolve, importing this module raises.  Build it with ``python -m esvit_amd.build``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESVIT_HIP_LIB") or os.path.join(_HERE, "lib", "libesvit_hip.so")  # (the override: A/B builds of tools/)

F32, BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_QGELU, EPI_QGELU_BWD = 0, 1, 2, 3, 4
GEMM_AUTO, GEMM_REGSTAGE, GEMM_DMA4, GEMM_DMA8, GEMM_DMA4W, GEMM_P8 = 0, 1, 2, 3, 4, 5

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("a_kstrided", i32), ("b_kstrided", i32),
        ("batch", i32),
        ("strideA", i64), ("strideB", i64), ("strideC", i64),
        ("bias", vp), ("residual", vp), ("ldr", i64),
        ("rowmap", vp), ("rowmap_period", i32), ("rowmap_tokens", i32),
        ("rowscale", vp), ("rows_per_sample", i32),
        ("aux", vp), ("ldaux", i64),
        ("epilogue", i32), ("out_f32", i32), ("splitk", i32),
        ("partial", vp), ("accumulate", i32), ("alpha", f32),
        ("colsum", vp), ("colsum_partial", vp),
        ("kernel", i32),
        ("rowstat", vp), ("rowstat_center", vp), ("rowstat_scale", f32),
        ("colstat", vp),
    ]


# name -> (restype, argtypes); every symbol declared in include/esvit_hip.h
SIGNATURES = {
    "esvit_version": (C.c_int, []),
    "esvit_last_error": (C.c_char_p, []),
    "esvit_relative_position_index": (C.c_int, [C.c_int, vp]),
    "esvit_window_maps": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "esvit_query": (i64, [C.c_int, i64, i64, i64]),
    "esvit_shift_mask": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esvit_gemm": (C.c_int, [C.c_int, C.POINTER(GemmDesc), vp]),
    "esvit_gemm_select": (C.c_int, [C.c_int, C.POINTER(GemmDesc), vp, vp, vp]),
    "esvit_mlp_fused_fwd": (C.c_int, [C.c_int, vp, vp, vp, f32, vp, vp, vp, vp, vp, i64, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "esvit_mlp_fused_fwd_train": (C.c_int, [C.c_int, vp, vp, vp, f32, vp, vp, vp, vp, vp, i64, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "esvit_mlp_fused_bwd": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, i64, C.c_int, vp, vp, vp, vp, vp, vp]),
    "esvit_ln_fold_finish": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "esvit_mlp_fused_weight": (C.c_int, [C.c_int, vp, vp, C.c_int, vp]),
    "esvit_cast_weight": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "esvit_layernorm_fwd": (C.c_int, [C.c_int, vp, vp, vp, f32, i64, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "esvit_conv_im2col": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, vp, vp]),
    "esvit_conv_col2im": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    vp, vp]),
    "esvit_dwconv3x3": (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esvit_dwconv3x3_wgrad": (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "esvit_col_sums2": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, vp, vp, vp]),
    "esvit_col_affine2": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, vp, vp, vp, C.c_int, vp, vp]),
    "esvit_pad_crop_tokens": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esvit_bn_fwd_coeffs": (C.c_int, [vp, f32, vp, vp, f32, f32, vp, vp, C.c_int, vp, vp]),
    "esvit_bn_eval_coeffs": (C.c_int, [vp, vp, vp, vp, f32, C.c_int, vp, vp]),
    "esvit_bn_bwd_local": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "esvit_bn_bwd_coeffs": (C.c_int, [vp, f32, vp, vp, C.c_int, vp, vp]),
    "esvit_layernorm_bwd": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, i64, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp]),
    "esvit_gather_cast": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "esvit_cast_f32_to": (C.c_int, [C.c_int, vp, vp, i64, vp]),
    "esvit_colsum": (C.c_int, [C.c_int, vp, i64, C.c_int, i64, vp, vp, C.c_int, vp]),
    "esvit_patch_im2col": (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "esvit_merge_ln_fwd": (C.c_int, [C.c_int, vp, vp, vp, f32, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "esvit_merge_ln_bwd": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "esvit_token_mean_fwd": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "esvit_token_mean_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esvit_attn_branch_fwd": (C.c_int, [C.c_int, vp, vp, vp, f32, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32,
                                        vp, vp, vp, vp, vp, vp, vp, vp]),
    "esvit_window_attn_fwd": (C.c_int, [C.c_int, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, vp]),
    "esvit_window_attn_bwd": (C.c_int, [C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, vp]),
    "esvit_relpos_bias_bwd": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "esvit_l2norm_fwd": (C.c_int, [C.c_int, vp, i64, C.c_int, vp, vp, vp]),
    "esvit_l2norm_bwd": (C.c_int, [C.c_int, vp, vp, vp, i64, C.c_int, vp, vp]),
    "esvit_weightnorm_fwd": (C.c_int, [C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "esvit_weightnorm_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "esvit_aug_crops": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "esvit_heads_split": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esvit_heads_merge": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "esvit_softmax_rows_fwd": (C.c_int, [C.c_int, vp, i64, C.c_int, C.c_int, f32, vp]),
    "esvit_softmax_rows_bwd": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, C.c_int, f32, vp]),
    "esvit_softmax_rows_chunked_fwd": (C.c_int, [C.c_int, vp, i64, C.c_int, C.c_int, f32, vp, C.c_int, C.c_int, vp]),
    "esvit_softmax_rows_chunked_bwd": (C.c_int, [C.c_int, vp, vp, i64, C.c_int, C.c_int, f32, vp, C.c_int, C.c_int, vp]),
    "esvit_teacher_row_stats": (C.c_int, [C.c_int, vp, vp, f32, i64, C.c_int, vp, vp, vp]),
    "esvit_region_match": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "esvit_dino_ce_fwd_bwd": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, f32, f32, i64, C.c_int, vp, vp, vp, vp, vp, vp]),
    "esvit_rowstat_combine": (C.c_int, [vp, i64, C.c_int, vp, vp, vp]),
    "esvit_sum_f32": (C.c_int, [vp, i64, vp, vp]),
    "esvit_scale_inplace": (C.c_int, [C.c_int, vp, i64, vp, vp]),
    "esvit_center_ema": (C.c_int, [vp, vp, f32, f32, C.c_int, vp]),
    "esvit_grad_sqnorm": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp]),
    "esvit_fused_clip_update_ema": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp, f32, f32, f32, f32, f32, f32, f32, vp, vp]),
}


def _open():
    lib = C.CDLL(LIB_PATH)
    try:
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
    except AttributeError:
        import _ctypes
        _ctypes.dlclose(lib._handle)  # a stale build: unload it, or the rebuilt file would resolve to this mapping again
        raise
    return lib


def _load():
    """Load the library; if it is missing or older than this binding (a symbol of include/esvit_hip.h is absent),
    (re)build it in-tree with hipcc once.  There is no non-HIP fallback: failure to build or load raises."""
    try:
        if os.path.exists(LIB_PATH):
            return _open()
    except (AttributeError, OSError):
        pass
    import importlib.util
    spec = importlib.util.spec_from_file_location("_esvit_amd_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build(force=False, verbose=False)
    return _open()


lib = _load()


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.esvit_last_error().decode()))
