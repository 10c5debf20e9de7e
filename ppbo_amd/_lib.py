"""ctypes binding of libppbo_hip.so (the C-ABI declared in include/ppbo_hip.h).

There is NO CPU fallback: if the shared library is missing or a symbol is
absent this module raises.  The oracle under oracle/ is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own libamdhip64.so; it must be in the process BEFORE libppbo_hip.so is
# dlopen'ed so both resolve to the same HIP runtime (device pointers are per-runtime).
import torch  # noqa: F401  (plumbing: device memory and streams)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libppbo_hip.so")

PPBO_ERR_NOT_PD = 1001
KERNEL_IDS = {"SE_kernel": 0, "RQ_kernel": 1, "camphor_copper_kernel": 2}
SCORE_MEAN, SCORE_POINTWISE_EI, SCORE_VARIANCE = 0, 1, 2


class FitOpts(C.Structure):
    _fields_ = [("gtol", C.c_double), ("maxiter", C.c_int), ("verbose", C.c_int), ("initial_radius", C.c_double),
                ("lbfgs_max_evals", C.c_int), ("judge_by_gradient_below_noise", C.c_int),
                ("start_is_whitened", C.c_int)]


class FitStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("n_cholesky", C.c_int), ("converged", C.c_int),
                ("T", C.c_double), ("gradnorm", C.c_double),
                ("lbfgs_iterations", C.c_int), ("lbfgs_evals", C.c_int), ("lbfgs_status", C.c_int)]


class Model(C.Structure):
    _fields_ = [("kernel_id", C.c_int), ("N", C.c_int), ("D", C.c_int), ("m", C.c_int),
                ("theta", C.c_double * 3), ("d_X", C.c_void_p), ("d_alpha", C.c_void_p),
                ("d_lam_diag", C.c_void_p), ("d_lam_off", C.c_void_p), ("d_G", C.c_void_p), ("kstar_fp32", C.c_int),
                ("d_Gt", C.c_void_p)]


_vp, _i, _d, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_int64
_dp3 = C.POINTER(C.c_double)

# symbol -> argtypes; every entry of include/ppbo_hip.h must be here (tests check the header against this)
SIGNATURES = {
    "ppbo_abi_version": [],
    "ppbo_ctx_create": [_i, C.POINTER(_vp)],
    "ppbo_ctx_destroy": [_vp],
    "ppbo_last_error": [_vp, C.c_char_p, C.c_size_t],
    "ppbo_profile_enable": [_vp, _i],
    "ppbo_profile_reset": [_vp],
    "ppbo_profile_read": [_vp, C.c_char_p, C.POINTER(_d), C.POINTER(_i)],
    "ppbo_gram": [_vp, _i, _vp, _i, _i, _dp3, _d, _vp, _vp],
    "ppbo_regularize_covariance": [_vp, _vp, _i, _i, _d, _i, _d, _vp],
    "ppbo_store_floor": [_vp, _vp, _i, _vp],
    "ppbo_cross_cov": [_vp, _i, _vp, _i, _vp, _i, _i, _dp3, _vp, _i, _vp],
    "ppbo_potrf": [_vp, _vp, _i, _i, C.POINTER(_i), _vp],
    "ppbo_pd_inverse": [_vp, _vp, _i, _vp, C.POINTER(_i), _vp],
    "ppbo_pd_inverse_factors": [_vp, _vp, _i, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_pd_inverse_ex": [_vp, _vp, _i, _vp, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_pd_inverse_append": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_pd_inverse_append_ex": [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_laplace_terms": [_vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp],
    "ppbo_sum_phi": [_vp, _vp, _i, _i, _d, _i, _vp, _vp],
    "ppbo_fit_fmap": [_vp, _vp, _i, _i, _d, _vp, C.POINTER(FitOpts), _vp, C.POINTER(FitStats), _vp],
    "ppbo_fit_fmap_whitened": [_vp, _vp, _i, _vp, _i, _i, _d, _vp, C.POINTER(FitOpts), _vp, C.POINTER(FitStats), _vp],
    "ppbo_gp_fit": [_vp, _i, _vp, _i, _i, _dp3, _d, _i, _vp, C.POINTER(FitOpts), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                    C.POINTER(FitStats), C.POINTER(_i), _vp],
    "ppbo_T_and_grad": [_vp, _vp, _vp, _i, _i, _d, C.POINTER(_d), _vp, _vp],
    "ppbo_posterior": [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_predict": [_vp, C.POINTER(Model), _vp, _i64, _i, _d, _vp, _vp, _vp, C.POINTER(_d), C.POINTER(_i64), _vp],
    "ppbo_predict_record": [_vp, C.POINTER(Model), _vp, _i64, _i, _d, _i64, _vp, _vp],
    "ppbo_predict_cov": [_vp, C.POINTER(Model), _vp, _i, _d, _vp, _vp, _vp],
    "ppbo_transposed_G_shape": [_i, C.POINTER(_i), C.POINTER(_i)],
    "ppbo_transposed_G": [_vp, _vp, _i, _vp, _vp],
    "ppbo_mean_grad": [_vp, C.POINTER(Model), _vp, C.c_int64, _vp, _vp, _vp],
    "ppbo_mean_search": [_vp, C.POINTER(Model), _vp, _i64, _i, _d, _i, _d, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_mean_ascent": [_vp, C.POINTER(Model), _vp, _i, _i, _d, _vp, _vp, _vp, _vp],
    "ppbo_shift_points": [_vp, _vp, _i64, _i, C.POINTER(_d), _vp, _vp],
    "ppbo_mean_search_multi": [_vp, C.POINTER(Model), _vp, _i64, C.POINTER(_d), _i, _vp, _i, C.POINTER(_d), _i, _d, _i, _d, _i, _vp, _vp, _vp],
    "ppbo_line_acq": [_vp, C.POINTER(Model), _vp, _i, _i, _d, _vp, _i, _d, _d, _vp, _vp, _vp],
    "ppbo_line_acq_xi": [_vp, C.POINTER(Model), _vp, _vp, _vp, _i, _i, _i, _d, _vp, _i, _d, _d, _vp, _vp, _vp],
    "ppbo_randn": [_vp, C.c_uint64, _vp, _i64, _vp],
    "ppbo_rff_project": [_vp, _vp, _i, _i, _vp, _i, _vp, _d, _vp, _vp],
    "ppbo_rff_score": [_vp, _vp, _i64, _i, _vp, _i, _vp, _d, _vp, _vp, C.POINTER(_d), C.POINTER(_i64), _vp],
    "ppbo_rff_search": [_vp, _vp, _i64, _i, _vp, _i, _vp, _d, _vp, _i, _d, _i, _d, _vp, _vp, C.POINTER(_i), _vp],
    "ppbo_rff_terms": [_vp, _vp, _i, _i, _i, _d, _vp, C.POINTER(_d), _vp, _vp, _vp],
    "ppbo_rff_omega_map": [_vp, _vp, _i, _i, _i, _d, _vp, _i, _d, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i), _vp],
    "ppbo_lu_slogdet": [_vp, _vp, _i, _i, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i), _vp],
    "ppbo_laplace_logdet": [_vp, _vp, _vp, _vp, _i, _i, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i), _vp],
    "ppbo_dgemv": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "ppbo_dist_unique_id": [_vp, _vp],
    "ppbo_dist_init": [_vp, _vp, _i, _i],
    "ppbo_dist_destroy": [_vp],
    "ppbo_argmax_allgather": [_vp, _d, _i64, C.POINTER(_d), C.POINTER(_i64), _vp],
    "ppbo_argmax_allgather_record": [_vp, _vp, C.POINTER(_d), C.POINTER(_i64), _vp],
    "ppbo_search_sharded": [_vp, C.POINTER(Model), _vp, _i64, _i, _d, _i64, C.POINTER(_d), C.POINTER(_i64), _vp],
    "ppbo_argmax_combine": [_vp, _vp, _i, C.POINTER(_d), C.POINTER(_i64), _vp],
    "ppbo_dgemm": [_vp, _i, _i, _i, _i, _i, _d, _vp, _i, _vp, _i, _d, _vp, _i, _vp],
}

_lib = None
ABI_VERSION = 6     # must equal PPBO_ABI_VERSION of include/ppbo_hip.h


def _check_stamp():
    """The .so is git-ignored: make sure it was built from the csrc/ + header that are on disk now."""
    from . import build as _build
    if os.environ.get("PPBO_SKIP_STAMP_CHECK"):
        return
    if not os.path.exists(_build.STAMP):
        raise ImportError(f"{_build.STAMP} is missing: {LIB_PATH} is of unknown provenance; rebuild it with "
                          "`python -m ppbo_amd.build --force`")
    if open(_build.STAMP).read().strip() != _build._digest():
        raise ImportError(f"{LIB_PATH} is stale (csrc/ or include/ppbo_hip.h changed since it was built): "
                          "rebuild it with `python -m ppbo_amd.build`")


def load():
    """Load the shared library and bind every declared symbol; raises loudly on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m ppbo_amd.build` (hipcc, gfx950). "
            "ppbo_amd has no CPU fallback.")
    _check_stamp()
    lib = C.CDLL(LIB_PATH)
    lib.ppbo_abi_version.restype = C.c_int
    if lib.ppbo_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} reports ABI {lib.ppbo_abi_version()}, this binding expects {ABI_VERSION}: "
                          "rebuild it with `python -m ppbo_amd.build --force`")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError(f"{LIB_PATH} does not export {name}")
        fn.argtypes = argtypes
        fn.restype = C.c_int
    _lib = lib
    return lib
