"""ctypes binding of libautoprog_hip.so (the C ABI declared in include/autoprog_hip.h).

The library is the product: there is NO fallback.  Importing this module on a machine
without the built .so, or calling an op without a GPU tensor, raises immediately.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded BEFORE the extension: both must share torch's HIP runtime)
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AP_LIB_PATH") or os.path.join(_HERE, "libautoprog_hip.so")      # AP_LIB_PATH: ablation builds (tools/abl_tn.sh)


class GemmEpilogue(Structure):
    _fields_ = [("bias", c_void_p), ("gelu", c_int), ("preact_out", c_void_p), ("dgelu_of", c_void_p),
                ("row_scale", c_void_p), ("rows_per_scale", c_int), ("residual", c_void_p), ("ldr", c_int), ("mul_by", c_void_p),
                ("mul_by8", c_void_p), ("q8_out", c_void_p), ("q8_scale", c_void_p), ("q8_amax", c_void_p)]


class MlpFusedArgs(Structure):
    """ap_mlp_fused_args (include/autoprog_hip.h)"""
    _fields_ = [("x", c_void_p), ("ldx", c_int), ("wa", c_void_p), ("ldwa", c_int), ("wb", c_void_p), ("ldwb", c_int), ("out", c_void_p), ("ldo", c_int),
                ("hidden_out", c_void_p), ("ldh", c_int), ("codes", c_void_p), ("bias1", c_void_p), ("bias2", c_void_p),
                ("row_scale_hidden", c_void_p), ("row_scale_out", c_void_p), ("rows_per_scale", c_int), ("residual", c_void_p), ("ldr", c_int),
                ("m", c_int), ("c", c_int), ("hidden", c_int), ("backward", c_int),
                ("ln_in", c_void_p), ("ld_ln", c_int), ("ln_out", c_void_p), ("ld_lno", c_int), ("ln_gamma", c_void_p), ("ln_beta", c_void_p),
                ("ln_eps", c_float), ("ln_mean", c_void_p), ("ln_rstd", c_void_p)]


class TnProblem(Structure):
    """ap_tn_problem (include/autoprog_hip.h)"""
    _fields_ = [("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int), ("C", c_void_p), ("ldc", c_int),
                ("M", c_int), ("N1", c_int), ("N2", c_int), ("alpha", c_float), ("colsum_A", c_void_p), ("colsum_weight", c_void_p), ("colsum_scale", c_float), ("b_patch", c_void_p), ("b_bn", c_void_p)]


class BnInput(Structure):
    """ap_bn_input (include/autoprog_hip.h)"""
    _fields_ = [("mean", c_void_p), ("rstd", c_void_p), ("gamma", c_void_p), ("beta", c_void_p)]


class LnReduce(Structure):
    """ap_ln_reduce (include/autoprog_hip.h)"""
    _fields_ = [("partial", c_void_p), ("n_partial", c_int), ("C", c_int), ("dgamma", c_void_p), ("dbeta", c_void_p)]


class PatchMap(Structure):
    """ap_patch_map (include/autoprog_hip.h)"""
    _fields_ = [("group", c_int), ("group_stride", c_int), ("row_stride", c_int), ("kseg", c_int), ("kseg_stride", c_int)]


TN_MAX_GROUP = 32          # AP_TN_MAX_GROUP
TN_MAX_GROUP_DET = 8       # the deterministic mode (workspace) takes 8
LN_MAX_BATCH = 12          # AP_LN_MAX_BATCH
_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
_SIGNATURES = {
    "ap_abi_version": (c_int, []),
    "ap_error_string": (c_char_p, [_I]),
    "ap_cast_f32_bf16": (_I, [_P, _P, _L, _P]),
    "ap_cast_bf16_f32": (_I, [_P, _P, _L, _P]),
    "ap_cast_transpose_f32_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "ap_resize_bilinear_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ap_droppath_masks": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ap_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "ap_layernorm_fwd_fp8": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "ap_layernorm_bwd_workspace": (ctypes.c_size_t, [_L, _I]),
    "ap_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, _P]),
    "ap_layernorm_bwd_partial_pool": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, ctypes.c_size_t, POINTER(c_int), _P]),
    "ap_gemm_nt": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, POINTER(GemmEpilogue), _P]),
    "ap_gemm_tn_acc": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "ap_gemm_tn_grouped_workspace": (ctypes.c_size_t, [_P, _I]),
    "ap_gemm_tn_acc_grouped": (_I, [_P, _I, _P, ctypes.c_size_t, _P]),
    "ap_gemm_tn_acc_grouped_ln": (_I, [_P, _I, _P, _I, _P, ctypes.c_size_t, _P]),
    "ap_colsum_acc": (_I, [_P, _I, _P, _I, _I, _P]),
    "ap_outlook_fwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ap_outlook_bwd": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "ap_avgpool2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ap_avgpool2_bwd_acc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ap_mhsa_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P]),
    "ap_mhsa_bwd_workspace": (ctypes.c_size_t, [_I, _I, _I, _I]),
    "ap_mhsa_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, ctypes.c_size_t, _P]),
    "ap_class_attn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ap_class_attn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "ap_mix_token_swap": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ap_soft_ce_fwd_bwd": (_I, [_P, _I, _P, _L, _L, _L, _I, _P, _P, _F, _L, _I, _F, _I, _P]),
    "ap_soft_ce_sparse_fwd_bwd": (_I, [_P, _I, _P, _P, _I, _L, _L, _I, _F, _P, _P, _F, _L, _I, _F, _I, _P]),
    "ap_loss_combine": (_I, [_P, _L, _F, _P, _L, _F, _P, _P]),
    "ap_row_scale": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "ap_add_bcast": (_I, [_P, _P, _P, _L, _L, _P]),
    "ap_sum_reps_acc": (_I, [_P, _L, _L, _I, _P]),
    "ap_resample_grid": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ap_bn_relu_workspace": (ctypes.c_size_t, [_L, _I]),
    "ap_bn_relu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, _P]),
    "ap_bn_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, _P]),
    "ap_sumsq_workspace": (ctypes.c_size_t, []),
    "ap_sumsq_f32": (_I, [_P, _L, _P, _P, ctypes.c_size_t, _P]),
    "ap_adamw_ema_step": (_I, [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P, _F, _F, _P, POINTER(c_void_p), POINTER(c_float), _I, _P, _P]),
    "ap_mix_token_swap_dev": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "ap_soft_ce_fwd_bwd_dev": (_I, [_P, _I, _P, _L, _L, _L, _I, _P, _P, _F, _L, _I, _F, _I, _P, _P]),
    "ap_soft_ce_sparse_fwd_bwd_dev": (_I, [_P, _I, _P, _P, _I, _L, _L, _I, _F, _P, _P, _F, _L, _I, _F, _I, _P, _P]),
    "ap_batched_transpose_bf16": (_I, [_P, _P, _P, _I, _I, _P]),
}
_SIGNATURES["ap_resize_bilinear_s2d16"] = (_I, [_P, _P, _I, _I, _I, _I, _I, _P])
_SIGNATURES["ap_conv7_pack"] = (_I, [_P, _P, _P])
_SIGNATURES["ap_conv7_s2d_stat_rows"] = (_I, [_I, _I, _I])
_SIGNATURES["ap_conv7_s2d"] = (_I, [_P, _P, _P, _I, _I, _I, _P, _P])
_SIGNATURES["ap_conv7_s2d_wgrad_workspace"] = (ctypes.c_size_t, [_I, _I, _I])
_SIGNATURES["ap_conv7_s2d_wgrad"] = (_I, [_P, _P, _P, _I, _I, _I, _P, ctypes.c_size_t, _P])
_SIGNATURES["ap_layernorm_bwd_partial"] = (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, POINTER(c_int), _P])
_SIGNATURES["ap_layernorm_bwd_reduce_batched"] = (_I, [_P, _I, _P])
_SIGNATURES["ap_quantize_fp8"] = (_I, [_P, _P, _L, _P, _P, _P])
_SIGNATURES["ap_quantize_fp8_multi"] = (_I, [_P, _I, _P, _P, _P])
_SIGNATURES["ap_debug_poison_lds"] = (_I, [ctypes.c_uint, _P, _P])
_SIGNATURES["ap_conv3x3_c64_bn"] = (_I, [_P, POINTER(BnInput), _P, _P, _I, _I, _I, _P, _P])
_SIGNATURES["ap_conv3x3_c64_wgrad_bn"] = (_I, [_P, POINTER(BnInput), _P, _P, _I, _I, _I, _P, ctypes.c_size_t, _P])
_SIGNATURES["ap_conv3x3_c64_bwd_stats"] = (_I, [_P, _P, _P, _I, _I, _I, _P, POINTER(BnInput), _P, _P])
_SIGNATURES["ap_bn_relu_bwd_act"] = (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, _P])
_SIGNATURES["ap_bn_relu_bwd_partials"] = (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _L, _I, _P, ctypes.c_size_t, _P])
_SIGNATURES["ap_mhsa_fwd_fp8"] = (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P])
_SIGNATURES["ap_gemm_nt_fp8"] = (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, POINTER(GemmEpilogue), _P])
_SIGNATURES["ap_gemm_nt_patch"] = (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, POINTER(PatchMap), _I, _P])
_SIGNATURES["ap_gemm_nt_patch_bn"] = (_I, [_P, POINTER(BnInput), _P, _I, _P, _I, _I, _I, _I, _P, POINTER(PatchMap), _I, _P])
_SIGNATURES["ap_bn_relu_fwd_partials"] = (_I, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _L, _I, _P])
_SIGNATURES["ap_conv3x3_c64_stat_rows"] = (_I, [_I, _I, _I])
_SIGNATURES["ap_conv3x3_c64_pack"] = (_I, [_P, _P, _P, _P])
_SIGNATURES["ap_conv3x3_c64"] = (_I, [_P, _P, _P, _I, _I, _I, _P, _P])
_SIGNATURES["ap_conv3x3_c64_wgrad_workspace"] = (ctypes.c_size_t, [_I, _I, _I])
_SIGNATURES["ap_conv3x3_c64_wgrad"] = (_I, [_P, _P, _P, _I, _I, _I, _P, ctypes.c_size_t, _P])
_SIGNATURES["ap_conv7_s2d_ld"] = (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P])
_SIGNATURES["ap_conv7_s2d_wgrad_ld"] = (_I, [_P, _P, _I, _P, _I, _I, _I, _P, ctypes.c_size_t, _P])
_SIGNATURES["ap_conv3x3_c128_pack"] = (_I, [_P, _P, _P, _P])
_SIGNATURES["ap_conv3x3_c128_stat_rows"] = (_I, [_I, _I, _I])
_SIGNATURES["ap_conv3x3_c128"] = (_I, [_P, _P, _P, _I, _I, _I, _P, _P])
_SIGNATURES["ap_conv3x3_c128_wgrad_workspace"] = (ctypes.c_size_t, [_I, _I, _I])
_SIGNATURES["ap_conv3x3_c128_wgrad"] = (_I, [_P, _P, _P, _I, _I, _I, _P, ctypes.c_size_t, _P])
# ap_sum_reps_acc(x, out, n, reps, stream)
_SIGNATURES["ap_sum_reps_acc"] = (_I, [_P, _P, _L, _I, _P])

_SIGNATURES["ap_mlp_fused"] = (_I, [POINTER(MlpFusedArgs), _P])
_SIGNATURES["ap_calib_copy"] = (_I, [_P, _P, _L, _P])
_SIGNATURES["ap_calib_mfma"] = (_I, [_P, _P, _I, _P])

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())
EXPECTED_ABI = 7                     # ap_abi_version() of the library these ctypes Structures mirror (include/autoprog_hip.h)


class AutoProgHipError(RuntimeError):
    pass


def _load():
    if not os.path.isfile(LIB_PATH):
        raise AutoProgHipError(
            "libautoprog_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C autoprog_amd/csrc`; there is no CPU/PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.ap_abi_version.restype = ctypes.c_int
    got = lib.ap_abi_version()
    if got != EXPECTED_ABI:          # (a stale or ablation .so with every symbol but older structs would read garbage past their end)
        raise AutoProgHipError("%s has ABI version %d, these bindings were written for %d: rebuild it (make -C autoprog_amd/csrc)"
                               % (LIB_PATH, got, EXPECTED_ABI))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code, what):
    if code != 0:
        raise AutoProgHipError("%s failed: %s (code %d)" % (what, lib.ap_error_string(code).decode(), code))
