"""ctypes binding of libgingr_hip.so (the C ABI declared in include/gingr_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, an exception is raised.
The library is built in-tree by ``__graft_entry__.build()`` (``make -C gingr_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GINGR_HIP_LIB: A/B timing of two builds on one box (development aid); default = the in-tree library
LIB_PATH = os.environ.get("GINGR_HIP_LIB") or os.path.join(_HERE, "libgingr_hip.so")

GINGR_OK = 0
ERR_BAD_ARGUMENT, ERR_HIP, ERR_NONFINITE, ERR_NOT_SPD, ERR_NO_DEVICE, ERR_STATE = 1, 2, 3, 4, 5, 6
NUM_PHASES = 3
NUM_SEGMENTS = 2
SEGMENT_FULLFIT = 2   # the gather of a sharded surface update (gingr_hip.h)
SEGMENT_REVSUM = 3    # per-template-vertex sums of the sharded reversed correspondence direction
FLAVOUR_CPD, FLAVOUR_ICP, FLAVOUR_ICP_SURFACE = 0, 1, 2
RCCL_UNIQUE_ID_BYTES = 128

_STATUS_NAMES = {
    1: "GINGR_ERR_BAD_ARGUMENT", 2: "GINGR_ERR_HIP", 3: "GINGR_ERR_NONFINITE", 4: "GINGR_ERR_NOT_SPD",
    5: "GINGR_ERR_NO_DEVICE", 6: "GINGR_ERR_STATE",
}


class GingrNativeError(RuntimeError):
    def __init__(self, code: int, where: str, text: str = ""):
        self.code = code
        super().__init__(f"{where}: {_STATUS_NAMES.get(code, code)} {text}".strip())


class StateScalars(ctypes.Structure):
    """gingr_state_scalars"""
    _fields_ = [
        ("euler", c_double * 3), ("center", c_double * 3), ("translation", c_double * 3),
        ("scale", c_double), ("sigma2", c_double), ("iteration", c_int32), ("status", c_int32),
    ]


class CpdParams(ctypes.Structure):
    _fields_ = [("w", c_double), ("lambda_", c_double)]


class IcpParams(ctypes.Structure):
    _fields_ = [("initial_sigma", c_double), ("end_sigma", c_double), ("max_iterations", c_int32)]


class MhRequest(ctypes.Structure):
    """gingr_mh_request"""
    _fields_ = [
        ("flavour", c_int32), ("kind", c_int32), ("cpd", POINTER(CpdParams)), ("icp", POINTER(IcpParams)),
        ("z", POINTER(c_double)), ("alpha", POINTER(c_double)), ("scalars", POINTER(StateScalars)),
        ("eval_sdev", c_double), ("eval_points", c_int64), ("need_forward", c_int32),
    ]


class MhResult(ctypes.Structure):
    """gingr_mh_result"""
    _fields_ = [
        ("scalars", StateScalars), ("log_value", c_double), ("dist_sum", c_double), ("dist_max", c_double), ("count", c_int64),
        ("log_q_forward", c_double), ("log_q_backward", c_double), ("forward_status", c_int32), ("backward_status", c_int32),
    ]


# int (*gingr_allreduce_fn)(void *user, int32_t segment, void *device_ptr, int64_t count)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int32, c_void_p, c_int64)
_dp = POINTER(c_double)
_ip = POINTER(c_int32)

KERNEL_GAUSSIAN_MIXTURE, KERNEL_DOT, KERNEL_LOOKUP = 0, 1, 2
OPT_CULL, OPT_FINE_CULL, OPT_NN_GRID, OPT_TRI_GRID, OPT_SPLIT_EXCHANGE, OPT_GRAM_DOWNDATE = 0, 1, 2, 3, 4, 5  # gingr_ctx_option


class ScalarKernel(ctypes.Structure):
    """gingr_scalar_kernel"""
    _fields_ = [("kind", c_int32), ("n_kernels", c_int32), ("sigmas", _dp), ("scalings", _dp), ("mirror", c_double),
                ("scaling", c_double), ("lookup", _dp)]


# name -> (restype, argtypes); every symbol include/gingr_hip.h declares
SIGNATURES = {
    "gingr_device_count": (c_int, []),
    "gingr_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "gingr_ctx_destroy": (None, [c_void_p]),
    "gingr_last_error": (c_char_p, [c_void_p]),
    "gingr_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "gingr_ctx_get_stream": (c_void_p, [c_void_p]),
    "gingr_ctx_synchronize": (c_int, [c_void_p]),
    "gingr_build_info": (c_char_p, []),
    "gingr_ctx_set_option": (c_int, [c_void_p, c_int32, c_int32]),
    "gingr_ctx_get_option": (c_int, [c_void_p, c_int32, POINTER(c_int32)]),
    "gingr_cpd_stats": (c_int, [c_void_p, c_int64, _dp, c_int64, _dp, c_double, c_double, _dp, _dp, _dp, _dp, _dp]),
    "gingr_cpd_initial_sigma2": (c_int, [c_void_p, c_int64, _dp, c_int64, _dp, _dp]),
    "gingr_nn": (c_int, [c_void_p, c_int64, _dp, c_int64, _dp, _ip, _dp, _dp]),
    "gingr_gauss_block": (c_int, [c_void_p, c_int64, _dp, c_int64, _dp, c_double, c_double, _dp]),
    "gingr_model_upload": (c_int, [c_void_p, c_int64, c_int32, _dp, _dp, _dp, _dp, c_int64, c_int64, POINTER(c_void_p)]),
    "gingr_model_destroy": (None, [c_void_p]),
    "gingr_model_gram_exchange": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "gingr_model_finalize": (c_int, [c_void_p, c_void_p]),
    "gingr_fitter_set_meshes": (c_int, [c_void_p, c_int64, POINTER(c_int32), c_int64, POINTER(c_int32)]),
    "gingr_fitter_set_surface_method": (c_int, [c_void_p, c_int32]),
    "gingr_fitter_set_correspondence_direction": (c_int, [c_void_p, c_int32]),
    "gingr_fitter_get_reversed_correspondence": (c_int, [c_void_p, POINTER(c_int32), _dp]),
    "gingr_fitter_update_icp_surface_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32]),
    "gingr_fitter_set_stop_threshold": (c_int, [c_void_p, c_double]),
    "gingr_fitter_stop_rule_hit": (c_int, [c_void_p, POINTER(c_int32)]),
    "gingr_fitter_icp_surface_phase_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32]),
    "gingr_fitter_get_surface_correspondence": (c_int, [c_void_p, _dp, _dp]),
    "gingr_fitter_surface_distance_stats": (c_int, [c_void_p, c_int32, c_int64, _dp, c_int32, c_double, _dp]),
    "gingr_fitter_mh_step": (c_int, [c_void_p, POINTER(MhRequest), _dp, _dp, POINTER(MhResult)]),
    "gingr_fitter_mh_restore": (c_int, [c_void_p]),
    "gingr_classic_cpd_create": (c_int, [c_void_p, c_int32, c_int64, _dp, c_int64, _dp, c_double, c_double, c_double, POINTER(c_void_p)]),
    "gingr_classic_cpd_destroy": (None, [c_void_p]),
    "gingr_classic_cpd_iterate": (c_int, [c_void_p, c_int32]),
    "gingr_classic_cpd_get": (c_int, [c_void_p, _dp, POINTER(c_double), _dp, _dp]),
    "gingr_classic_cpd_set": (c_int, [c_void_p, _dp, c_double]),
    "gingr_mesh_distance_stats": (c_int, [c_void_p, c_int64, _dp, c_int64, _dp, c_int64, POINTER(c_int32), c_int32, c_double, _dp]),
    "gingr_fitter_update_icp_surface_sample_async": (c_int, [c_void_p, POINTER(IcpParams), _dp]),
    "gingr_fitter_posterior_logpdf_icp_surface": (c_int, [c_void_p, POINTER(IcpParams), _dp, POINTER(c_double)]),
    "gingr_rigid_icp_create": (c_int, [c_void_p, c_int32, c_int64, _dp, c_int64, _dp, POINTER(c_void_p)]),
    "gingr_rigid_icp_destroy": (None, [c_void_p]),
    "gingr_rigid_icp_iterate": (c_int, [c_void_p, c_int32, _dp]),
    "gingr_rigid_icp_get": (c_int, [c_void_p, _dp, _dp]),
    "gingr_rigid_icp_set": (c_int, [c_void_p, _dp]),
    "gingr_fitter_set_fit_points": (c_int, [c_void_p, _dp]),
    "gingr_nicp_solve": (c_int, [c_void_p, c_int32, c_int64, _dp, c_int64, _ip, _dp, _dp, c_int32, _ip, _dp, c_double, c_double, c_double,
                                 _dp, _dp]),
    "gingr_mesh_closest_points": (c_int, [c_void_p, c_int64, _dp, c_int64, _dp, c_int64, _ip, _dp, _dp, _ip, _dp]),
    "gingr_model_new_reference": (c_int, [c_void_p, c_void_p, c_int64, _dp, _ip, _dp, c_int64, c_int64, POINTER(c_void_p)]),
    "gingr_gpmm_build_diagonal": (c_int, [c_void_p, c_int64, _dp, POINTER(ScalarKernel), POINTER(ScalarKernel), POINTER(ScalarKernel),
                                          c_double, c_int32, c_int64, c_int64, POINTER(c_void_p)]),
    "gingr_gpmm_build_gaussian": (c_int, [c_void_p, c_int64, _dp, c_int32, _dp, _dp, c_double, c_int32, c_int64, c_int64,
                                          POINTER(c_void_p)]),
    "gingr_pointset_distance_extrema": (c_int, [c_void_p, _dp, c_int64, _dp, _dp]),
    "gingr_model_download": (c_int, [c_void_p, c_void_p, _dp, _dp, _dp, _dp]),
    "gingr_model_num_points": (c_int64, [c_void_p]),
    "gingr_model_rank": (c_int32, [c_void_p]),
    "gingr_model_instance": (c_int, [c_void_p, c_void_p, _dp, _dp, _dp, _dp, c_double, _dp]),
    "gingr_model_coefficients": (c_int, [c_void_p, c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "gingr_model_posterior_mean": (c_int, [c_void_p, c_void_p, _dp, _dp, _dp, _dp, _dp, c_int32, _ip, _dp, _dp, _dp, _dp]),
    "gingr_fitter_create": (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    "gingr_fitter_destroy": (None, [c_void_p]),
    "gingr_fitter_set_target": (c_int, [c_void_p, c_int64, _dp]),
    "gingr_fitter_set_landmarks": (c_int, [c_void_p, c_int32, _ip, _dp, _dp]),
    "gingr_fitter_set_options": (c_int, [c_void_p, c_int32, c_double]),
    "gingr_fitter_set_state": (c_int, [c_void_p, _dp, POINTER(StateScalars)]),
    "gingr_fitter_get_state": (c_int, [c_void_p, _dp, POINTER(StateScalars), _dp]),
    "gingr_fitter_get_cpd_stats": (c_int, [c_void_p, _dp, _dp, _dp, _dp]),
    "gingr_fitter_get_icp_idx": (c_int, [c_void_p, _ip, _dp]),
    "gingr_fitter_update_cpd_async": (c_int, [c_void_p, POINTER(CpdParams), c_int32]),
    "gingr_fitter_update_icp_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32]),
    "gingr_fitter_update_cpd_sample_async": (c_int, [c_void_p, POINTER(CpdParams), _dp]),
    "gingr_fitter_update_icp_sample_async": (c_int, [c_void_p, POINTER(IcpParams), _dp]),
    "gingr_fitter_posterior_logpdf_cpd": (c_int, [c_void_p, POINTER(CpdParams), _dp, _dp]),
    "gingr_fitter_posterior_logpdf_icp": (c_int, [c_void_p, POINTER(IcpParams), _dp, _dp]),
    "gingr_fitter_retry_counter": (c_int, [c_void_p, c_int32, POINTER(c_int32)]),
    "gingr_fitter_exchange": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64)]),
    "gingr_fitter_cpd_phase_async": (c_int, [c_void_p, POINTER(CpdParams), c_int32]),
    "gingr_fitter_update_cpd_sharded_async": (c_int, [c_void_p, POINTER(CpdParams), c_int32, ALLREDUCE_FN, c_void_p]),
    "gingr_fitter_update_icp_sharded_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32, ALLREDUCE_FN, c_void_p]),
    "gingr_fitter_icp_phase_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32]),
    "gingr_fitter_update_sharded_async": (c_int, [c_void_p, c_int32, POINTER(CpdParams), POINTER(IcpParams), c_int32, _dp, ALLREDUCE_FN, c_void_p]),
    "gingr_fitter_posterior_logpdf_sharded": (c_int, [c_void_p, c_int32, POINTER(CpdParams), POINTER(IcpParams), _dp, ALLREDUCE_FN, c_void_p, _dp]),
    "gingr_fitter_fullfit_exchange": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "gingr_fitter_reversal_exchange": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "gingr_fitter_gather_stage": (c_int, [c_void_p, c_int32, c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64)]),
    "gingr_fitter_gather_finish": (c_int, [c_void_p, c_int32]),
    "gingr_fitter_update_rccl_async": (c_int, [c_void_p, c_int32, POINTER(CpdParams), POINTER(IcpParams), c_int32, _dp]),
    "gingr_fitter_posterior_logpdf_rccl": (c_int, [c_void_p, c_int32, POINTER(CpdParams), POINTER(IcpParams), _dp, _dp]),
    "gingr_group_set_meshes": (c_int, [c_void_p, c_int64, _ip, c_int64, _ip]),
    "gingr_group_set_surface_method": (c_int, [c_void_p, c_int32]),
    "gingr_group_set_correspondence_direction": (c_int, [c_void_p, c_int32]),
    "gingr_group_update_async": (c_int, [c_void_p, c_int32, POINTER(CpdParams), POINTER(IcpParams), c_int32, _dp]),
    "gingr_group_posterior_logpdf": (c_int, [c_void_p, c_int32, POINTER(CpdParams), POINTER(IcpParams), _dp, _dp]),
    "gingr_rccl_load": (c_int, [c_void_p, c_char_p]),
    "gingr_rccl_unique_id": (c_int, [c_void_p, c_void_p]),
    "gingr_ctx_rccl_init": (c_int, [c_void_p, c_void_p, c_int32, c_int32]),
    "gingr_ctx_rccl_info": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), c_char_p, c_int32]),
    "gingr_ctx_rccl_allreduce_async": (c_int, [c_void_p, c_void_p, c_int64]),
    "gingr_fitter_update_cpd_rccl_async": (c_int, [c_void_p, POINTER(CpdParams), c_int32]),
    "gingr_fitter_update_icp_rccl_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32]),
    "gingr_group_create": (c_int, [c_int32, _ip, POINTER(c_void_p)]),
    "gingr_group_destroy": (None, [c_void_p]),
    "gingr_group_size": (c_int32, [c_void_p]),
    "gingr_group_last_error": (c_char_p, [c_void_p]),
    "gingr_group_ctx": (c_void_p, [c_void_p, c_int32]),
    "gingr_group_shard_rows": (c_int, [c_void_p, c_int32, POINTER(c_int64), POINTER(c_int64)]),
    "gingr_group_model_upload": (c_int, [c_void_p, c_int64, c_int32, _dp, _dp, _dp, _dp]),
    "gingr_group_gpmm_build_gaussian": (c_int, [c_void_p, c_int64, _dp, c_int32, _dp, _dp, c_double, c_int32]),
    "gingr_group_model_rank": (c_int32, [c_void_p]),
    "gingr_group_set_target": (c_int, [c_void_p, c_int64, _dp]),
    "gingr_group_set_landmarks": (c_int, [c_void_p, c_int32, _ip, _dp, _dp]),
    "gingr_group_set_options": (c_int, [c_void_p, c_int32, c_double]),
    "gingr_group_set_state": (c_int, [c_void_p, _dp, POINTER(StateScalars)]),
    "gingr_group_get_state": (c_int, [c_void_p, _dp, POINTER(StateScalars), _dp]),
    "gingr_group_update_cpd_async": (c_int, [c_void_p, POINTER(CpdParams), c_int32]),
    "gingr_group_update_icp_async": (c_int, [c_void_p, POINTER(IcpParams), c_int32]),
    "gingr_group_synchronize": (c_int, [c_void_p]),
    "gingr_group_exchange_info": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "gingr_ctx_nn_counting": (c_int, [c_void_p, c_int32]),
    "gingr_ctx_nn_tests": (c_int, [c_void_p, POINTER(c_int64)]),
    "gingr_ctx_timing_enable": (c_int, [c_void_p, c_int32]),
    "gingr_ctx_timing_read": (c_int, [c_void_p, c_int32, _dp, POINTER(c_int64)]),
    "gingr_ctx_timing_reset": (c_int, [c_void_p]),
}

_LIB = None


def load():
    """Load libgingr_hip.so and set the prototypes.  Raises if the library has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GingrNativeError(
                ERR_STATE, "load",
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        lib = ctypes.CDLL(LIB_PATH)
        # GINGR_HIP_LIB_ALLOW_OLDER=1 (set by tools/abn.sh only: same-box A/B timing against an OLDER build of the library)
        # tolerates symbols that build does not have yet and says which; any other library -- the in-tree one, or a user's
        # GINGR_HIP_LIB override -- must export every declared symbol
        older_build = os.environ.get("GINGR_HIP_LIB_ALLOW_OLDER") == "1"
        skipped = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            except AttributeError:
                if older_build:
                    skipped.append(name)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        if skipped:
            import sys
            print(f"gingr_amd: {LIB_PATH} lacks {len(skipped)} declared symbols (older build): {', '.join(skipped)}", file=sys.stderr)
        _LIB = lib
    return _LIB


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)
