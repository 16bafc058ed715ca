"""ctypes binding of libsmcmi.so (include/smcmi.h).

The library is the product: if it is missing or cannot be loaded this module raises - there is no
Python / CPU fallback for any of the engine's entry points.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMCMI_LIBRARY selects another build of the same library (tests: libsmcmi_strict.so, every floating-point contraction off)
LIB_PATH = os.environ.get("SMCMI_LIBRARY") or os.path.join(os.path.dirname(_HERE), "csrc", "libsmcmi.so")

MAX_PARA = 64
MAX_CAND = 16

PRIOR = {"normal": 0, "uniform": 1, "gamma": 2, "beta": 3, "invgamma": 4, "rootinvgamma": 5}
LIK = {"none": -1, "gauss_iso": 0, "linreg": 1, "linmodel3": 2, "capm_literal": 3, "lgss_kalman": 4, "host_callback": 100}
RESAMPLE = {"systematic": 0, "multinomial": 1, "polyalgo": 1}

ERRORS = {-1: "ARG", -2: "HIP", -3: "NAN_ESS", -4: "POSDEF", -5: "CAPACITY", -6: "BRACKET", -7: "UNSUPPORTED", -8: "STATE", -9: "CALLBACK", -10: "TIMEOUT"}

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
lp = C.POINTER(C.c_int64)


class Config(C.Structure):
    _fields_ = [("n_parts", C.c_int64), ("n_local", C.c_int64), ("gid0", C.c_int64), ("n_para", C.c_int32),
                ("device", C.c_int32), ("seed", C.c_uint64), ("max_stages", C.c_int32), ("store_history", C.c_int32)]


class RunConfig(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("n_mh_steps", C.c_int32), ("lam", C.c_double), ("n_phi", C.c_int32),
                ("resampling_method", C.c_int32), ("threshold_ratio", C.c_double), ("c", C.c_double),
                ("alpha", C.c_double), ("target", C.c_double), ("use_fixed_schedule", C.c_int32),
                ("tempering_target", C.c_double), ("tempered_update_prior_weight", C.c_double),
                ("log_prob_old_data", C.c_double), ("solver_passes", C.c_int32), ("sync_every", C.c_int32),
                ("use_graph", C.c_int32), ("initial_ess", C.c_double), ("phi_rtol", C.c_double),
                ("stop_after_stage", C.c_int32), ("continue_run", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("resamples", C.c_int32), ("logmdd", C.c_double), ("c", C.c_double),
                ("accept", C.c_double), ("seconds", C.c_double), ("kernel_ms_mutate", C.c_double),
                ("n_mutate_launches", C.c_int32), ("solver_passes", C.c_int64), ("solver_stalls", C.c_int32),
                ("select_stalls", C.c_int32), ("spec_stalls", C.c_int32), ("paused", C.c_int32),
                ("n_segments", C.c_int32), ("segment_stages", C.c_int32), ("kernel_ms_segments", C.c_double),
                ("segment_blocks", C.c_int32), ("segment_state", C.c_int32), ("segment_timeouts", C.c_int32),
                ("shift_fallback_stage", C.c_int32)]


class LoopState(C.Structure):
    _fields_ = [("stage_index", C.c_int32), ("j", C.c_int32), ("resampled_last_period", C.c_int32), ("resamples", C.c_int32),
                ("phi_n", C.c_double), ("phi_prop", C.c_double), ("c", C.c_double), ("accept", C.c_double),
                ("ess", C.c_double), ("logmdd", C.c_double)]


class StageStats(C.Structure):
    _fields_ = [("ess", C.c_double), ("sum_unnorm", C.c_double), ("logz_inc", C.c_double), ("resample", C.c_int32)]


# int (*smcmi_lik_callback)(const double *theta, int64_t m, int64_t d, double *out, void *user_data)
LIK_CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_void_p)

# smcmi_host_comm: the collectives of a sharded run as host functions (include/smcmi.h)
HC_ALLGATHER = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_void_p)
HC_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                           C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)
HC_BARRIER = C.CFUNCTYPE(C.c_int, C.c_void_p)


class HostComm(C.Structure):
    _fields_ = [("allgather", HC_ALLGATHER), ("alltoallv", HC_ALLTOALLV), ("barrier", HC_BARRIER), ("user", C.c_void_p)]


# every symbol include/smcmi.h declares: (name, restype, argtypes)
_H = C.c_void_p
SYMBOLS = [
    ("smcmi_create", C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    ("smcmi_destroy", C.c_int, [_H]),
    ("smcmi_last_error", C.c_char_p, []),
    ("smcmi_version", C.c_int, []),
    ("smcmi_set_parameters", C.c_int, [_H, ip, dp, dp, ip, dp, dp]),
    ("smcmi_set_likelihood", C.c_int, [_H, C.c_int32, C.c_int32, dp, C.c_int64, dp, C.c_int64, C.c_int64, dp, C.c_int64, C.c_int64]),
    ("smcmi_set_likelihood_callback", C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p]),
    ("smcmi_eval_cloud_callback", C.c_int, [_H, C.c_int32, C.c_int32]),
    ("smcmi_callback_stats", C.c_int, [_H, lp, lp]),
    ("smcmi_callback_phases", C.c_int, [_H, dp, C.c_int32]),
    ("smcmi_upload_cloud", C.c_int, [_H, dp]),
    ("smcmi_download_cloud", C.c_int, [_H, dp]),
    ("smcmi_upload_cloud_device", C.c_int, [_H, C.c_void_p]),
    ("smcmi_init_from_prior", C.c_int, [_H]),
    ("smcmi_initialize_likelihoods", C.c_int, [_H]),
    ("smcmi_bridge_resample", C.c_int, [_H, _H, C.c_int32, C.c_uint32, C.c_int64, dp, lp]),
    ("smcmi_copy_rows", C.c_int, [_H, C.c_int64, _H, C.c_int64, C.c_int64]),
    ("smcmi_normalize_weights", C.c_int, [_H, C.c_int32]),
    ("smcmi_cloud_device_ptr", C.c_int, [_H, C.POINTER(C.c_void_p), lp]),
    ("smcmi_ess_at", C.c_int, [_H, dp, C.c_int32, C.c_double, dp]),
    ("smcmi_solve_phi", C.c_int, [_H, dp, C.c_int32, ip, dp, C.c_double, C.c_double, C.c_double, ip, dp]),
    ("smcmi_correct", C.c_int, [_H, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(StageStats)]),
    ("smcmi_resample", C.c_int, [_H, C.c_int32, C.c_uint32, dp, lp]),
    ("smcmi_moments", C.c_int, [_H, dp, dp]),
    ("smcmi_mutate", C.c_int, [_H, dp, dp, ip, ip, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_uint32, dp]),
    ("smcmi_propose", C.c_int, [_H, dp, dp, ip, ip, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint32, dp, dp, dp]),
    ("smcmi_accept", C.c_int, [_H, dp, dp, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32]),
    ("smcmi_run", C.c_int, [_H, C.POINTER(RunConfig), C.POINTER(Result)]),
    ("smcmi_stages_held", C.c_int, [_H, ip]),
    ("smcmi_get_stage_records", C.c_int, [_H, dp, dp, dp, dp, ip]),
    ("smcmi_get_history", C.c_int, [_H, dp, dp]),
    ("smcmi_get_loop_state", C.c_int, [_H, C.POINTER(LoopState)]),
    ("smcmi_set_loop_state", C.c_int, [_H, C.POINTER(LoopState)]),
    ("smcmi_set_stage_records", C.c_int, [_H, C.c_int32, dp, dp, dp, dp, ip]),
    ("smcmi_set_history", C.c_int, [_H, C.c_int32, dp, dp]),
    ("smcmi_comm_buffer", C.c_int, [_H, C.POINTER(C.c_void_p), lp]),
    ("smcmi_comm_read", C.c_int, [_H, dp, C.c_int64]),
    ("smcmi_shard_ess_partial", C.c_int, [_H, dp, C.c_int32, C.c_double]),
    ("smcmi_shard_correct_partial", C.c_int, [_H, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32]),
    ("smcmi_shard_normalize_moments_partial", C.c_int, [_H, C.c_double, C.c_int32, dp, C.c_int32]),
    ("smcmi_shard_weights_device_ptr", C.c_int, [_H, C.POINTER(C.c_void_p)]),
    ("smcmi_shard_resample", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, lp]),
    ("smcmi_shard_mutate_partial", C.c_int, [_H, dp, dp, ip, ip, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_uint32]),
    ("smcmi_sync", C.c_int, [_H]),
    ("smcmi_debug_proposal_densities", C.c_int, [dp, dp, dp, dp, C.c_int32, C.c_double, C.c_double, dp, dp]),
    ("smcmi_comm_unique_id", C.c_int, [C.c_char_p]),
    ("smcmi_comm_init", C.c_int, [_H, C.c_int32, C.c_int32, C.c_char_p]),
    ("smcmi_run_sharded", C.c_int, [_H, C.POINTER(RunConfig), C.POINTER(Result)]),
    ("smcmi_comm_init_host", C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(HostComm)]),
    ("smcmi_mailbox_export", C.c_int, [_H, C.POINTER(C.c_uint8)]),
    ("smcmi_mailbox_import", C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    ("smcmi_mailbox_selftest", C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("smcmi_mailbox_active", C.c_int, [_H, C.POINTER(C.c_int32)]),
    ("smcmi_run_group", C.c_int, [C.POINTER(_H), C.c_int32, C.POINTER(RunConfig), C.POINTER(Result)]),
]

_LIB = None


class SMCMIError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("smcmi error %s (%d): %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process do not share devices, so
    when torch is installed (it is the plumbing for torch.distributed / device tensors in the multi-GPU path) load ITS
    runtime first; libsmcmi.so's NEEDED libamdhip64.so.7 then resolves to the same, already-loaded copy whatever the
    import order.  Without torch the system ROCm runtime is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("SMCMI_RCCL_PATH", rccl)      # the sharded driver dlopens the same RCCL copy torch uses
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libsmcmi.so (built in-tree by __graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libsmcmi.so not found at %s - build it with `python __graft_entry__.py` "
                              "(hipcc --offload-arch=gfx950); the engine has no CPU fallback" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise SMCMIError(rc, lib().smcmi_last_error().decode())
