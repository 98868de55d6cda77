"""ctypes binding of libproxgrad_hip.so (include/proxgrad_hip.h).  There is NO CPU fallback: if the
library is missing or a call fails, an exception is raised."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PG_LIB_PATH: another build of the SAME library (scripts/sanitize_host.sh loads the host-instrumented one); never a fallback
LIB_PATH = os.environ.get("PG_LIB_PATH") or os.path.join(HERE, "libproxgrad_hip.so")

PG_F32, PG_F64 = 0, 1
PG_G_ZERO, PG_G_NORML1, PG_G_INDBOX, PG_G_SQRNORML2 = 0, 1, 2, 3
PG_SEQ_ADAPTIVE, PG_SEQ_FIXED, PG_SEQ_SIMPLE, PG_SEQ_CONSTANT, PG_SEQ_HOST, PG_SEQ_REPEATED = 0, 1, 2, 3, 4, 5
PG_FLAG_GAMMA_TOO_SMALL, PG_FLAG_SWEEP_FALLBACK, PG_FLAG_COOP_SLOW = 1, 2, 4
PG_K_GEMV_N, PG_K_GEMV_N_FINISH, PG_K_GEMV_T, PG_K_EPILOGUE, PG_K_EXTRAPOLATE, PG_K_DR_STEP = range(6)
KERNEL_NAMES = ["gemv_n_partial", "gemv_n_finish", "gemv_t", "fb_epilogue", "extrapolate", "dr_step", "gemv_tn"]


PG_ERR_INVALID, PG_ERR_HIP, PG_ERR_ALLOC, PG_ERR_UNSUPPORTED, PG_ERR_COLLECTIVE, PG_ERR_TIMEOUT = -1, -2, -3, -4, -5, -6


class ProxGradError(RuntimeError):
    """A failed library call; ``code`` is the pg_status (PG_ERR_*; None for errors raised on the Python side).  Callers
    that fall back to another path do so on ``e.code == PG_ERR_UNSUPPORTED`` only -- a HIP or allocation failure
    propagates."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


class pg_device_info(C.Structure):
    _fields_ = [("device", C.c_int32), ("compute_units", C.c_int32), ("wavefront_size", C.c_int32),
                ("lds_bytes_per_cu", C.c_int32), ("global_mem_bytes", C.c_int64), ("clock_khz", C.c_int32),
                ("arch", C.c_char * 64), ("name", C.c_char * 128)]


class pg_iter_opts(C.Structure):
    _fields_ = [("fast", C.c_int32), ("adaptive", C.c_int32), ("Lf", C.c_double), ("gamma", C.c_double),
                ("minimum_gamma", C.c_double), ("reduce_gamma", C.c_double), ("increase_gamma", C.c_double),
                ("mf", C.c_double), ("seq_kind", C.c_int32), ("seq_p0", C.c_double), ("seq_p1", C.c_double),
                ("g_kind", C.c_int32), ("g_p0", C.c_double), ("g_p1", C.c_double), ("reuse_residual", C.c_int32),
                ("single_sweep", C.c_int32)]


class pg_iter_scalars(C.Structure):
    _fields_ = [("gamma", C.c_double), ("f_x", C.c_double), ("g_z", C.c_double), ("res_inf", C.c_double),
                ("beta", C.c_double), ("f_z", C.c_double), ("f_z_upp", C.c_double), ("n_backtracks", C.c_int32),
                ("flags", C.c_int32), ("a_passes", C.c_int64)]


class pg_iter_state(C.Structure):
    _fields_ = [("x", C.c_void_p), ("grad_f_x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p),
                ("res", C.c_void_p), ("z_prev", C.c_void_p), ("grad_f_z", C.c_void_p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
ALLREDUCE_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

_vp, _i32, _i64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_size_t
_pf64 = C.POINTER(C.c_double)

# name -> (argtypes); every function returns pg_status (int32) unless listed in _SPECIAL
SIGNATURES = {
    "pg_ctx_create": [_i32, _vp, C.POINTER(_vp)],
    "pg_ctx_destroy": [_vp],
    "pg_ctx_set_stream": [_vp, _vp],
    "pg_ctx_set_allreduce": [_vp, ALLREDUCE_FN, _vp],
    "pg_ctx_set_allreduce_async": [_vp, ALLREDUCE_FN, ALLREDUCE_WAIT_FN, _vp],
    "pg_comm_get_unique_id": [_vp],
    "pg_ctx_comm_init": [_vp, _vp, _i32, _i32, _i32],
    "pg_ctx_comm_destroy": [_vp],
    "pg_ctx_comm_stats": [_vp, C.POINTER(_i64), C.POINTER(_i64)],
    "pg_ctx_sync": [_vp],
    "pg_ctx_capture_begin": [_vp],
    "pg_ctx_capture_end": [_vp, C.POINTER(_vp)],
    "pg_graph_launch": [_vp],
    "pg_graph_destroy": [_vp],
    "pg_ctx_device_info": [_vp, C.POINTER(pg_device_info)],
    "pg_ctx_profile_enable": [_vp, _i32],
    "pg_ctx_profile_select": [_vp, C.c_uint32],
    "pg_ctx_set_column_sharding": [_vp, _i32, _i32],
    "pg_ctx_test_team_fault": [_vp, _i32, _i32],
    "pg_ctx_test_team_slack": [_vp, C.POINTER(_i64), C.POINTER(_i64)],
    "pg_ctx_row_team_tune": [_vp, C.c_char_p, _i64],
    "pg_ctx_row_team_geometry": [_vp, C.c_char_p, _i64],
    "pg_ctx_row_team_alloc": [_vp, C.POINTER(_vp), C.POINTER(_i64)],
    "pg_ctx_row_team_export": [_vp, _vp],
    "pg_ctx_row_team_import": [_vp, _vp, C.POINTER(_vp)],
    "pg_ctx_set_row_team": [_vp, _i32, _i32, C.POINTER(_vp), _i32],
    "pg_ctx_row_team_stats": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "pg_ctx_row_team_selftest": [_vp, _pf64],
    "pg_ctx_profile_reset": [_vp],
    "pg_ctx_profile_read": [_vp, _i32, C.POINTER(_i64), _pf64],
    "pg_malloc": [_vp, _sz, C.POINTER(_vp)],
    "pg_free": [_vp, _vp],
    "pg_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "pg_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "pg_memcpy_d2d": [_vp, _vp, _vp, _sz],
    "pg_memset_zero": [_vp, _vp, _sz],
    "pg_mat_create": [_vp, _i32, _i64, _i64, C.POINTER(_vp)],
    "pg_mat_destroy": [_vp],
    "pg_mat_upload": [_vp, _vp, _i64],
    "pg_mat_set_from_device": [_vp, _vp, _i64],
    "pg_mat_download": [_vp, _vp, _i64],
    "pg_mat_generate": [_vp, C.c_uint32, _i64, _f64],
    "pg_mat_generate_block": [_vp, C.c_uint32, _i64, _i64, _f64],
    "pg_mat_info": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_vp)],
    "pg_mat_rank1_update": [_vp, _f64, _vp, _vp],
    "pg_mat_mul": [_vp, _vp, _vp],
    "pg_mat_mul_multi": [_vp, _i32, C.POINTER(_vp), C.POINTER(_vp)],
    "pg_mat_mul_adjoint": [_vp, _vp, _vp],
    "pg_mat_fused_tn": [_vp, _vp, _vp, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _pf64],
    "pg_mat_fused_tn_res": [_vp, _vp, _vp, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _pf64],
    "pg_mat_fused_tn_pair": [_vp, _vp, _vp, _vp, _vp, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pf64],
    "pg_mat_fused_tn_pair_res": [_vp, _vp, _vp, _vp, _vp, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pf64],
    "pg_mat_fused_tn_trio": [_vp, _vp, _vp, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _i32, _pf64],
    "pg_mat_fused_dys": [_vp, _vp, _vp, _vp, _f64, _f64, _i32, _f64, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _pf64],
    "pg_ls_create": [_vp, _vp, _vp, _f64, C.POINTER(_vp)],
    "pg_ls_destroy": [_vp],
    "pg_ls_value_and_gradient": [_vp, _vp, _vp, _pf64],
    "pg_ls_value": [_vp, _vp, _pf64],
    "pg_ls_gradient": [_vp, _vp, _vp, _pf64],
    "pg_ls_residual_ptr": [_vp, C.POINTER(_vp)],
    "pg_ls_fused_pass": [_vp, _vp, _vp, _f64, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _pf64],
    "pg_prox_norml1": [_vp, _i32, _i64, _vp, _vp, _f64, _f64, _pf64],
    "pg_prox_indbox": [_vp, _i32, _i64, _vp, _vp, _f64, _f64, _vp, _vp, _pf64],
    "pg_norml1_value": [_vp, _i32, _i64, _vp, _f64, _pf64],
    "pg_prox_norml1w": [_vp, _i32, _i64, _vp, _vp, _vp, _f64, _pf64],
    "pg_norml1w_value": [_vp, _i32, _i64, _vp, _vp, _pf64],
    "pg_axpby": [_vp, _i32, _i64, _vp, _f64, _vp, _f64, _vp],
    "pg_add_scalar": [_vp, _i32, _i64, _vp, _vp, _f64],
    "pg_fill": [_vp, _i32, _i64, _vp, _f64],
    "pg_extrapolate": [_vp, _i32, _i64, _vp, _vp, _vp, _f64],
    "pg_dot": [_vp, _i32, _i64, _vp, _vp, _pf64],
    "pg_nrm2sq": [_vp, _i32, _i64, _vp, _pf64],
    "pg_nrminf": [_vp, _i32, _i64, _vp, _pf64],
    "pg_fb_epilogue": [_vp, _i32, _i64, _vp, _vp, _f64, _i32, _f64, _f64, _vp, _vp, _vp, _pf64],
    "pg_loss_value_and_gradient": [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _pf64],
    "pg_prox_sepquad": [_vp, _i32, _i64, _vp, _vp, _vp, _f64, _vp, _f64, _f64, _pf64],
    "pg_dr_step": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _f64, _i32, _f64, _f64, _f64, _pf64],
    "pg_dr_step_async": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _f64, _i32, _f64, _f64, _f64, _i32],
    "pg_dr_step_wait": [_vp, _i32, _pf64],
    "pg_dr_run": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _f64, _i32, _f64, _f64, _f64, _f64, _i64,
                  _i32, C.POINTER(_i64), _pf64],
    "pg_iter_opts_default": [C.POINTER(pg_iter_opts)],
    "pg_iter_create": [_vp, _vp, C.POINTER(pg_iter_opts), C.POINTER(_vp)],
    "pg_iter_destroy": [_vp],
    "pg_iter_set_g_vectors": [_vp, _vp, _vp],
    "pg_iter_init": [_vp, _vp, C.POINTER(pg_iter_scalars)],
    "pg_iter_step": [_vp, _f64, C.POINTER(pg_iter_scalars)],
    "pg_iter_run": [_vp, _i64, _i64, _f64, C.POINTER(_i64), C.POINTER(pg_iter_scalars)],
    "pg_iter_run_batched": [_vp, _i64, _i64, _f64, _i32, C.POINTER(_i64), C.POINTER(pg_iter_scalars)],
    "pg_iter_run_small": [_vp, _i64, _i64, _f64, C.POINTER(_i64), C.POINTER(pg_iter_scalars)],
    "pg_iter_run_coop": [_vp, _i64, _i64, _f64, _i32, C.POINTER(_i64), C.POINTER(pg_iter_scalars)],
    "pg_iter_state_view": [_vp, C.POINTER(pg_iter_state)],
    "pg_iter_state_bytes": [_vp, C.POINTER(_i64)],
    "pg_iter_state_download": [_vp, _vp, _i64],
    "pg_iter_state_upload": [_vp, _vp, _i64, C.POINTER(pg_iter_scalars)],
    "pg_lbfgs_create": [_vp, _i32, _i32, _i64, C.POINTER(_vp)],
    "pg_lbfgs_destroy": [_vp],
    "pg_lbfgs_update": [_vp, _vp, _vp],
    "pg_lbfgs_reset": [_vp],
    "pg_lbfgs_apply": [_vp, _vp, _vp],
    "pg_lbfgs_images_enable": [_vp, _i64],
    "pg_lbfgs_images_update": [_vp, _vp, _vp],
    "pg_lbfgs_images_apply": [_vp, _vp, _vp],
    "pg_lbfgs_images_ready": [_vp, C.POINTER(_i32)],
}
_SPECIAL = {"pg_abi_version": ([], C.c_int32), "pg_last_error": ([], C.c_char_p), "pg_comm_available": ([], C.c_int32)}

_lib = None
PG_ABI_VERSION = 4  # include/proxgrad_hip.h (tests/test_cpu_host.py compares the two)


def load():
    """Load the shared library (once).  torch is imported first so that the HIP runtime already mapped
    by torch (libamdhip64.so.7) is the one our DT_NEEDED entry resolves to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ProxGradError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
            "(or `python proximalalgorithms.jl_amd/_build.py`).  There is no CPU fallback.")
    import torch  # noqa: F401  (maps torch's HIP runtime before ours is resolved)

    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    try:  # before any other symbol is looked up: a stale build fails HERE, with its version, not at some later call
        lib.pg_abi_version.argtypes, lib.pg_abi_version.restype = [], C.c_int32
        found = int(lib.pg_abi_version())
    except AttributeError:
        found = None
    if found != PG_ABI_VERSION:
        raise ProxGradError(f"{LIB_PATH} has ABI version {found}, this package was written against {PG_ABI_VERSION}: rebuild it "
                            "(`python proximalalgorithms.jl_amd/_build.py --force`) or point PG_LIB_PATH at a matching build")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int32
    for name, (argtypes, restype) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def exported_symbols():
    return list(SIGNATURES) + list(_SPECIAL)


def check(status):
    if status != 0:
        msg = load().pg_last_error()
        raise ProxGradError(f"libproxgrad_hip error {status}: {msg.decode() if msg else '?'}", code=int(status))


def call(name, *args):
    check(getattr(load(), name)(*args))
