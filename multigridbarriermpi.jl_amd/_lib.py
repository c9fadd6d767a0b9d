"""ctypes binding of libmgb_hip.so (include/mgb_hip.h).  No torch types cross this boundary."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmgb_hip.so")

c_int_p = C.POINTER(C.c_int)
c_i32_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)
c_flt_p = C.POINTER(C.c_float)
c_ll_p = C.POINTER(C.c_longlong)
c_str_arr = C.POINTER(C.c_char_p)
H = C.c_void_p
# mgb_allreduce_fn: int (*)(void* user, double* dev_ptr, long long count); the pointer is passed as an integer
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong)


class MGBError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mgb error %d: %s" % (code, msg))
        self.code = code


# name -> argtypes (return type is always int unless listed in _SPECIAL)
PROTOTYPES = {
    "mgb_ctx_create": [C.c_int, C.POINTER(H)],
    "mgb_ctx_destroy": [H],
    "mgb_ctx_synchronize": [H],
    "mgb_ctx_set_comm": [H, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p],
    "mgb_rccl_unique_id": [C.c_char_p],
    "mgb_ctx_set_comm_rccl": [H, C.c_char_p, C.c_int, C.c_int],
    "mgb_ctx_comm_stats": [H, c_ll_p, c_dbl_p],
    "mgb_shard_rows": [C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p],
    "mgb_fem1d_native": [C.c_int, C.POINTER(H)],
    "mgb_fem2d_native": [C.c_int, c_dbl_p, C.c_int, C.POINTER(H)],
    "mgb_fem3d_native": [C.c_int, C.c_int, C.POINTER(H)],
    "mgb_geo_create": [C.c_int, C.c_int, C.c_int, C.c_int, c_dbl_p, c_dbl_p, C.POINTER(H)],
    "mgb_geo_set_matrix": [H, C.c_char_p, C.c_int, C.c_int, c_i32_p, c_i32_p, c_dbl_p],
    "mgb_geo_destroy": [H],
    "mgb_geo_dims": [H, c_int_p, c_int_p, c_int_p, c_int_p],
    "mgb_geo_get_xw": [H, c_dbl_p, c_dbl_p],
    "mgb_geo_matrix_info": [H, C.c_char_p, c_int_p, c_int_p, c_int_p],
    "mgb_geo_matrix_get": [H, C.c_char_p, c_i32_p, c_i32_p, c_dbl_p],
    "mgb_vec_create": [H, C.c_int, c_dbl_p, C.POINTER(H)],
    "mgb_vec_free": [H],
    "mgb_vec_len": [H, c_int_p],
    "mgb_vec_upload": [H, c_dbl_p],
    "mgb_vec_download": [H, c_dbl_p],
    "mgb_csr_create": [H, C.c_int, C.c_int, c_i32_p, c_i32_p, c_dbl_p, C.POINTER(H)],
    "mgb_csr_free": [H],
    "mgb_csr_dims": [H, c_int_p, c_int_p, c_int_p],
    "mgb_csr_get": [H, c_i32_p, c_i32_p, c_dbl_p],
    "mgb_csr_spgemm": [H, H, C.POINTER(H)],
    "mgb_csr_transpose": [H, C.POINTER(H)],
    "mgb_csr_add": [H, C.c_double, H, C.POINTER(H)],
    "mgb_csr_hcat": [C.c_int, C.POINTER(H), C.POINTER(H)],
    "mgb_csr_blockdiag": [C.c_int, C.POINTER(H), C.POINTER(H)],
    "mgb_diag": [H, H, C.c_int, C.c_int, C.POINTER(H)],
    "mgb_spmv": [H, H, H],
    "mgb_spmv_add": [H, H, H, H],
    "mgb_dot": [H, H, c_dbl_p],
    "mgb_norm": [H, c_dbl_p],
    "mgb_sum": [H, c_dbl_p],
    "mgb_col_extract": [H, C.c_int, C.c_int, C.c_int, H],
    "mgb_map_rows_barrier": [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p,
                             C.c_int, H, H],
    "mgb_mul": [H, H, H],
    "mgb_axpy": [H, C.c_double, H, H],
    "mgb_vec_allreduce_sum": [H],
    "mgb_all_isfinite": [H, c_int_p],
    "mgb_amg_create": [H, H, C.c_int, c_str_arr, C.c_int, c_str_arr, C.c_int, c_int_p, C.c_int, C.c_double,
                       C.POINTER(H)],
    "mgb_amg_create_cones": [H, H, C.c_int, c_str_arr, C.c_int, c_str_arr, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p,
                             c_dbl_p, C.POINTER(H)],
    "mgb_amg_create_terms": [H, H, C.c_int, c_str_arr, C.c_int, c_str_arr, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p,
                             c_dbl_p, c_dbl_p, c_dbl_p, C.POINTER(H)],
    "mgb_amg_set_exponents": [H, C.c_int, c_dbl_p],
    "mgb_amg_set_term_mask": [H, C.POINTER(C.c_ubyte)],
    "mgb_amg_destroy": [H],
    "mgb_amg_dims": [H, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p],
    "mgb_amg_local_rows": [H, c_int_p, c_int_p, c_int_p],
    "mgb_amg_prepare": [H, C.c_int],
    "mgb_amg_chol_info": [H, C.c_int, c_int_p, c_dbl_p, c_int_p],
    "mgb_amg_chol_values_local": [H, C.c_int, c_int_p],
    "mgb_amg_level_size": [H, C.c_int, c_int_p, c_int_p],
    "mgb_amg_hessian_pattern": [H, C.c_int, c_i32_p, c_i32_p],
    "mgb_amg_set_c": [H, c_dbl_p],
    "mgb_amg_set_z": [H, c_dbl_p],
    "mgb_amg_get_z": [H, c_dbl_p],
    "mgb_amg_apply_D": [H, C.c_int, c_dbl_p, c_dbl_p],
    "mgb_amg_f0": [H, C.c_int, c_dbl_p, C.c_double, c_dbl_p, c_dbl_p],
    "mgb_amg_f0_trial": [H, C.c_int, c_dbl_p, c_dbl_p, C.c_double, c_dbl_p],
    "mgb_amg_f1": [H, C.c_int, c_dbl_p, C.c_double, c_dbl_p],
    "mgb_amg_f2": [H, C.c_int, c_dbl_p, C.c_double, c_dbl_p],
    "mgb_amg_set_early_stop": [H, C.c_int],
    "mgb_amg_f0_f32": [H, C.c_int, c_flt_p, C.c_float, c_dbl_p],
    "mgb_amg_f1_f32": [H, C.c_int, c_flt_p, C.c_float, c_flt_p],
    "mgb_amg_f2_f32": [H, C.c_int, c_flt_p, C.c_float, c_flt_p],
    "mgb_amg_f1_template_f64": [H, C.c_int, c_dbl_p, C.c_double, c_dbl_p],
    "mgb_amg_f2_template_f64": [H, C.c_int, c_dbl_p, C.c_double, c_dbl_p],
    "mgb_amg_solve_linear": [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_amg_solve_linear_gpu": [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_amg_set_solver": [H, C.c_int],
    "mgb_amg_set_schedule": [H, C.c_int],
    "mgb_amg_set_stop_rule": [H, C.c_int],
    "mgb_amg_set_centering": [H, C.c_int],
    "mgb_amg_set_pcg": [H, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int],
    "mgb_amg_sol_pcg": [H, c_ll_p, c_dbl_p],
    "mgb_hessian_apply": [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, C.c_int],
    "mgb_smooth": [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, C.c_int, C.c_int, C.c_double, C.c_int, c_dbl_p],
    "mgb_prolong": [H, C.c_int, c_dbl_p, c_dbl_p],
    "mgb_restrict": [H, C.c_int, c_dbl_p, c_dbl_p],
    "mgb_amg_prolongation": [H, C.c_int, c_int_p, c_int_p, c_int_p, c_i32_p, c_i32_p, c_dbl_p],
    "mgb_amg_pcg_solve_linear": [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_int_p, c_dbl_p, c_int_p],
    "mgb_amg_mg_info": [H, C.c_int, c_int_p],
    "mgb_amg_time_mg_kernels": [H, C.c_int, C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_plan_prolongation": [H, H, c_int_p, c_int_p, c_int_p, c_i32_p, c_i32_p, c_dbl_p],
    "mgb_amg_solve": [H, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int],
    "mgb_amg_sol_info": [H, c_int_p, c_dbl_p, c_dbl_p, c_ll_p],
    "mgb_amg_sol_get": [H, c_ll_p, c_dbl_p, c_dbl_p],
    "mgb_amg_sol_kernels": [H, c_dbl_p, c_dbl_p, c_ll_p],
    "mgb_amg_time_kernels": [H, C.c_int, C.c_int, C.c_int, c_dbl_p, c_dbl_p],
    "mgb_plan_create": [H, C.c_int, c_str_arr, C.c_int, c_str_arr, C.c_int, c_int_p, C.c_int, C.c_int,
                        C.POINTER(H)],
    "mgb_reduction_scratch_doubles": [C.c_int, C.c_int, c_ll_p],
    "mgb_plan_destroy": [H],
    "mgb_plan_sizes": [H, c_int_p, c_int_p, c_int_p, c_int_p],
    "mgb_plan_pattern": [H, c_i32_p, c_i32_p],
    "mgb_plan_eval_host": [H, c_dbl_p, c_dbl_p],
    "mgb_plan_shard": [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), c_int_p, c_int_p],
    "mgb_plan_apply_B_host": [H, c_dbl_p, c_dbl_p],
    "mgb_plan_apply_BT_host": [H, c_dbl_p, c_dbl_p],
    "mgb_plan_chol_bench": [H, c_dbl_p, C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_plan_hostchol_create": [H, C.c_int, C.POINTER(H)],
    "mgb_hostchol_destroy": [H],
    "mgb_hostchol_info": [H, c_int_p, c_int_p, c_dbl_p],
    "mgb_hostchol_factor_solve": [H, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_hostchol_partition": [H, C.c_int, c_int_p, C.c_int, c_int_p, c_int_p],
    "mgb_hostchol_factor_solve_dist": [H, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_plan_hostchol_create_ranked": [H, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(H)],
    "mgb_hostchol_rank_aligned": [H, C.c_int, c_int_p, c_int_p],
    "mgb_hostchol_factor_solve_dist_local": [H, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p, c_dbl_p, c_dbl_p, c_dbl_p],
    "mgb_plan_chol_tree": [H, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p],
    "mgb_chol_selftest": [C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p],
}
_SPECIAL = {"mgb_last_error": ([], C.c_char_p), "mgb_version": ([], C.c_int), "mgb_device_count": ([], C.c_int)}

_lib = None


def load():
    """Load libmgb_hip.so; fail loudly if it has not been built (there is no Python/CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmgb_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or multigridbarriermpi.jl_amd/csrc/build.sh (expected at %s)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (args, res) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise MGBError(rc, lib.mgb_last_error().decode("utf-8", "replace"))
    return rc


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_dbl_p)


def iptr(a):
    return None if a is None else a.ctypes.data_as(c_i32_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def str_array(pairs):
    flat = []
    for a, b in pairs:
        flat += [str(a).encode(), str(b).encode()]
    return (C.c_char_p * len(flat))(*flat)
