"""ctypes binding of libmi_gp.so (include/mi_gp.h).  There is no CPU fallback: if the HIP library
is missing the import of the product path fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_gp.so")

MAX_KERN = 8
KERNEL_IDS = {"RBF": 0, "Matern52": 1, "Matern32": 2, "Exponential": 3, "RatQuad": 4}
OP_IDS = {"+": 0, "*": 1}


class MiGpConfig(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int),
        ("d", ctypes.c_int),
        ("nkern", ctypes.c_int),
        ("kernel_ids", ctypes.c_int * MAX_KERN),
        ("ops", ctypes.c_int * MAX_KERN),
        ("device", ctypes.c_int),
        ("panel_tiles", ctypes.c_int),
    ]


class MiGpBuffers(ctypes.Structure):
    _fields_ = [
        ("X_dev", ctypes.c_void_p),
        ("y_dev", ctypes.c_void_p),
        ("K_dev", ctypes.c_void_p),
        ("lda", ctypes.c_long),
        ("Z_dev", ctypes.c_void_p),
        ("W_dev", ctypes.c_void_p),
    ]


class MiGpBatchBuffers(ctypes.Structure):
    _fields_ = [
        ("K_dev", ctypes.c_void_p),
        ("Z_dev", ctypes.c_void_p),
        ("W_dev", ctypes.c_void_p),
        ("stride_k", ctypes.c_long),
        ("stride_zw", ctypes.c_long),
        ("count", ctypes.c_int),
    ]


class MiGpShardConfig(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int),
        ("d", ctypes.c_int),
        ("nkern", ctypes.c_int),
        ("kernel_ids", ctypes.c_int * MAX_KERN),
        ("ops", ctypes.c_int * MAX_KERN),
        ("panel_tiles", ctypes.c_int),
        ("world", ctypes.c_int),
        ("rank", ctypes.c_int),
        ("device", ctypes.c_int),
        ("X_dev", ctypes.c_void_p),
        ("y_dev", ctypes.c_void_p),
        ("K_dev", ctypes.c_void_p),
        ("ldk", ctypes.c_long),
        ("P_dev", ctypes.c_void_p * 2),
        ("ldp", ctypes.c_long),
        ("theta_dev", ctypes.c_void_p),
        ("info_dev", ctypes.c_void_p),
        ("out_dev", ctypes.c_void_p),
    ]


_lib = None


def load():
    """Load libmi_gp.so and declare every entry point of include/mi_gp.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C andvaranaut_amd/csrc` (there is no CPU fallback for the GP hot path)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cl, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    lib.mi_gp_last_global_error.restype = ctypes.c_char_p
    lib.mi_gp_last_error.restype = ctypes.c_char_p
    lib.mi_gp_last_error.argtypes = [vp]
    lib.mi_gp_create.argtypes = [ctypes.POINTER(MiGpConfig), ctypes.POINTER(vp)]
    lib.mi_gp_destroy.argtypes = [vp]
    lib.mi_gp_padded_n.argtypes = [vp]
    lib.mi_gp_padded_n.restype = cl
    lib.mi_gp_num_theta.argtypes = [vp]
    lib.mi_gp_stream.argtypes = [vp]
    lib.mi_gp_stream.restype = vp
    lib.mi_gp_set_data.argtypes = [vp, ctypes.POINTER(MiGpBuffers)]
    lib.mi_gp_lml.argtypes = [vp, dp, dp]
    lib.mi_gp_lml_parts.argtypes = [vp, dp, dp]
    lib.mi_gp_lml_grad.argtypes = [vp, dp, dp, dp]
    lib.mi_gp_set_batch.argtypes = [vp, ctypes.POINTER(MiGpBatchBuffers)]
    lib.mi_gp_lml_batch.argtypes = [vp, ci, dp, dp, ip]
    lib.mi_gp_lml_grad_batch.argtypes = [vp, ci, dp, dp, dp, ip]
    lib.mi_gp_alpha.argtypes = [vp, dp]
    lib.mi_gp_grad_x.argtypes = [vp, vp]
    lib.mi_gp_set_diag.argtypes = [vp, vp]
    lib.mi_gp_factor.argtypes = [vp, dp]
    lib.mi_gp_predict.argtypes = [vp, vp, ci, vp, cl, vp, vp, ci]
    lib.mi_gp_predict_u.argtypes = [vp, vp, ci, vp, cl, vp, vp, ci]
    lib.mi_gp_predict_grad.argtypes = [vp, vp, ci, vp, cl, vp, vp, ci, vp, vp]
    lib.mi_gp_set_option.argtypes = [vp, ci, ci]
    lib.mi_gp_get_option.argtypes = [vp, ci, ip]
    lib.mi_gp_set_profiling.argtypes = [vp, ci]
    lib.mi_gp_timers.argtypes = [vp, dp, ci]
    lib.mi_gp_assemble_block.argtypes = [ci, ci, ip, ip, vp, vp, ci, vp, ci, ci, ci, vp, cl, ci, ci, ci, vp]
    lib.mi_gp_chol_panel.argtypes = [vp, cl, ci, ci, vp, vp, ci, vp]
    lib.mi_gp_lml_partial.argtypes = [vp, cl, vp, ci, vp, vp]
    lib.mi_gp_gemm_f64.argtypes = [ci, ci, ci, ci, ci, cd, vp, cl, vp, cl, cd, vp, cl, ci, ci, ci, cl, cl, cl, vp]
    lib.mi_gp_gemm_f64_tuned.argtypes = [ci, ci, ci, ci, ci, cd, vp, cl, vp, cl, cd, vp, cl, ci, ci, ci, ci, ci, ci, vp]
    lib.mi_gp_gemm_nt_kseg.argtypes = [ci, ci, ci, cd, vp, cl, vp, cl, ci, cl, cd, vp, cl, ci, ci, vp]
    lib.mi_gp_trsm_block.argtypes = [vp, cl, vp, ci, ci, vp, cl, ci, vp]
    lib.mi_gp_trmv_upper.argtypes = [vp, cl, vp, ci, vp, vp]
    lib.mi_gp_grad_contract_block_scratch.argtypes = [ci, ci, ci, ci]
    lib.mi_gp_grad_contract_block_scratch.restype = cl
    lib.mi_gp_grad_contract_block.argtypes = [ci, ci, ip, ip, vp, vp, ci, vp, cl, ci, ci, ci, vp, vp, cl, vp, vp]
    lib.mi_gp_shard_create.argtypes = [ctypes.POINTER(MiGpShardConfig), ctypes.POINTER(vp)]
    lib.mi_gp_shard_destroy.argtypes = [vp]
    lib.mi_gp_shard_begin.argtypes = [vp, ci, vp, vp]
    lib.mi_gp_shard_step.argtypes = [vp, ci, vp, vp]
    lib.mi_gp_shard_finish.argtypes = [vp, vp, vp]
    lib.mi_gp_shard_wait_piece.argtypes = [vp, ci, vp]
    lib.mi_gp_shard_set_option.argtypes = [vp, ci, ci]
    lib.mi_gp_shard_times.argtypes = [vp, dp, ci]
    lib.mi_gp_shard_piece_times.argtypes = [vp, dp, ci]
    lib.mi_gp_shard_chain_stream.argtypes = [vp]
    lib.mi_gp_shard_last_error.argtypes = [vp]
    lib.mi_gp_shard_last_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        getattr(lib, name)  # raises AttributeError if a declared symbol is missing
    _lib = lib
    return lib


# every symbol include/mi_gp.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "mi_gp_last_global_error",
    "mi_gp_last_error",
    "mi_gp_create",
    "mi_gp_destroy",
    "mi_gp_padded_n",
    "mi_gp_num_theta",
    "mi_gp_stream",
    "mi_gp_set_data",
    "mi_gp_lml",
    "mi_gp_lml_parts",
    "mi_gp_lml_grad",
    "mi_gp_set_batch",
    "mi_gp_lml_batch",
    "mi_gp_lml_grad_batch",
    "mi_gp_alpha",
    "mi_gp_grad_x",
    "mi_gp_set_diag",
    "mi_gp_factor",
    "mi_gp_predict",
    "mi_gp_predict_u",
    "mi_gp_predict_grad",
    "mi_gp_set_option",
    "mi_gp_get_option",
    "mi_gp_set_profiling",
    "mi_gp_timers",
    "mi_gp_gemm_f64",
    "mi_gp_gemm_f64_tuned",
    "mi_gp_gemm_nt_kseg",
    "mi_gp_assemble_block",
    "mi_gp_chol_panel",
    "mi_gp_lml_partial",
    "mi_gp_trsm_block",
    "mi_gp_trmv_upper",
    "mi_gp_grad_contract_block_scratch",
    "mi_gp_grad_contract_block",
    "mi_gp_shard_create",
    "mi_gp_shard_destroy",
    "mi_gp_shard_begin",
    "mi_gp_shard_step",
    "mi_gp_shard_finish",
    "mi_gp_shard_wait_piece",
    "mi_gp_shard_set_option",
    "mi_gp_shard_times",
    "mi_gp_shard_piece_times",
    "mi_gp_shard_chain_stream",
    "mi_gp_shard_last_error",
]
