"""ctypes binding of the C-ABI in include/sfb.h (libsfb.so, built in-tree by csrc/Makefile).

The library is the product: there is no Python/CPU fallback.  If libsfb.so is missing the import
of this module fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFB_LIB_PATH") or os.path.join(_HERE, "libsfb.so")  # env: A/B builds only

SFB_OK, SFB_ERR_INVALID_ARG, SFB_ERR_UNSUPPORTED, SFB_ERR_HIP, SFB_ERR_NO_DEVICE = range(5)


class SfbError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("sfb status %d: %s" % (status, msg))
        self.status = status


class SfbQPParams(C.Structure):
    """sfb_qp_params == smooth::feedback::QPSolverParams (qp_solver.hpp:29-68)."""

    _fields_ = [
        ("alpha", C.c_float),
        ("rho", C.c_float),
        ("sigma", C.c_float),
        ("scaling", C.c_int32),
        ("eps_abs", C.c_float),
        ("eps_rel", C.c_float),
        ("eps_primal_inf", C.c_float),
        ("eps_dual_inf", C.c_float),
        ("max_iter", C.c_int64),
        ("max_time_ns", C.c_int64),
        ("stop_check_iter", C.c_uint32),
        ("polish", C.c_int32),
        ("polish_iter", C.c_uint32),
        ("delta", C.c_float),
        ("verbose", C.c_int32),
        ("reuse_factor", C.c_int32),
    ]


class SfbMPCLayout(C.Structure):
    """sfb_mpc_layout (include/sfb.h)."""

    _fields_ = [
        ("nx", C.c_int32), ("nu", C.c_int32), ("ncr", C.c_int32), ("kmesh", C.c_int32), ("nivals", C.c_int32),
        ("tf", C.c_double), ("alpha", C.c_void_p), ("D", C.c_void_p), ("nparts", C.c_int32),
        ("part_kind", C.c_void_p), ("part_dof", C.c_void_p), ("crl", C.c_void_p), ("cru", C.c_void_p),
        ("jac_keep", C.c_void_p),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsfb.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C smooth_feedback_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # torch (when present) bundles its own libamdhip64.so.7; import it first so that exactly one HIP
    # runtime is mapped into the process and device pointers from torch tensors are valid here.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    L = C.CDLL(LIB_PATH)
    vp, dp, i64, i32 = C.c_void_p, C.c_void_p, C.c_int64, C.c_int
    L.sfb_version.restype = C.c_char_p
    L.sfb_last_error.restype = C.c_char_p
    L.sfb_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    L.sfb_device_count.argtypes = [C.POINTER(C.c_int)]
    L.sfb_qp_params_default.argtypes = [C.POINTER(SfbQPParams)]
    L.sfb_qp_params_default.restype = None
    L.sfb_qp_dense_solve_batch.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12 + [vp]
    L.sfb_qp_dense_solve_batch_host.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12
    L.sfb_qp_dense_solve_batch_host_multi.argtypes = L.sfb_qp_dense_solve_batch_host.argtypes
    L.sfb_qp_dense_solve_batch_host_trace.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12 + [dp, i32]
    L.sfb_qp_dense_solve_batch_trace.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12 + [dp, i32, vp]
    L.sfb_qp_dense_solve_batch_host_phases.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12 + [dp, i32, dp]
    L.sfb_qp_dense_solve_batch_phases.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12 + [dp, i32, dp, vp]
    L.sfb_qp_dense_solve_batch_ws.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32] + [dp] * 12 + [vp, vp]
    L.sfb_qp_dense_workspace_bytes.argtypes = [C.POINTER(SfbQPParams), i64, i32, i32, C.POINTER(C.c_int64)]
    L.sfb_workspace_create.argtypes = [i64, C.POINTER(C.c_void_p)]
    L.sfb_workspace_destroy.argtypes = [C.c_void_p]
    L.sfb_workspace_destroy.restype = None
    L.sfb_workspace_info.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    i32p = C.c_void_p
    L.sfb_sparse_qp_plan_create.argtypes = [i32, i32, i32p, i32p, i32p, i32p, i32, i32p, C.POINTER(C.c_void_p)]
    L.sfb_sparse_qp_plan_create_staged.argtypes = [i32, i32, i32p, i32p, i32p, i32p, i32, i32p, i32p, C.POINTER(C.c_void_p)]
    L.sfb_sparse_qp_plan_create_pruned.argtypes = [i32, i32, i32p, i32p, i32p, i32p, i32, i32p, i32p, C.c_void_p,
                                                   C.POINTER(C.c_void_p)]
    L.sfb_sparse_qp_plan_workspace_bytes.argtypes = [C.c_void_p, i64, C.POINTER(C.c_int64)]
    L.sfb_sparse_qp_plan_pruned_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sfb_sparse_qp_plan_destroy.argtypes = [C.c_void_p]
    L.sfb_sparse_qp_plan_destroy.restype = None
    L.sfb_sparse_qp_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sfb_sparse_qp_plan_get_perm.argtypes = [C.c_void_p, i32p]
    L.sfb_sparse_qp_plan_get_factor_order.argtypes = [C.c_void_p, i32, i32p]
    L.sfb_sparse_qp_solve_batch.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12 + [vp, vp]
    L.sfb_sparse_qp_solve_batch_ordered.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12 + [vp, vp, vp]
    L.sfb_sparse_qp_solve_batch_host.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12
    L.sfb_sparse_qp_solve_batch_host_multi.argtypes = L.sfb_sparse_qp_solve_batch_host.argtypes
    L.sfb_sparse_qp_solve_batch_host_trace.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12 + [dp, i32]
    L.sfb_sparse_qp_solve_batch_trace.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12 + [vp, dp, i32, vp]
    L.sfb_sparse_qp_solve_batch_host_phases.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12 + [dp, i32, dp]
    L.sfb_sparse_qp_solve_batch_phases.argtypes = [C.c_void_p, C.POINTER(SfbQPParams), i64] + [dp] * 12 + [vp, dp, i32, dp, vp]
    L.sfb_ekf_predict_batch.argtypes = [i64, i32, dp, dp, i32, dp, i32, dp, vp]
    L.sfb_ekf_predict_stepper_batch.argtypes = [i32, i64, i32, dp, dp, i32, dp, i32, dp, vp]
    L.sfb_ekf_predict_stepper_batch_host.argtypes = [i32, i64, i32, dp, dp, i32, dp, i32, dp]
    L.sfb_ekf_predict_rk4_batch.argtypes = [i64, i32, dp, dp, dp, dp, i32, dp, i32, dp, vp]
    L.sfb_ekf_predict_rk4_batch_host.argtypes = [i64, i32, dp, dp, dp, dp, i32, dp, i32, dp]
    L.sfb_ekf_update_batch.argtypes = [i64, i32, i32, dp, dp, i32, dp, dp, dp, dp, vp]
    L.sfb_ekf_predict_update_batch.argtypes = [i64, i32, i32, dp, dp, i32, dp, i32, dp, dp, i32, dp, dp, dp, dp, vp]
    L.sfb_ekf_step_batch_host.argtypes = [i64, i32, i32, dp, dp, i32, dp, i32, dp, dp, i32, dp, dp, dp, dp]
    lay = C.POINTER(SfbMPCLayout)
    L.sfb_mpc_record_doubles.argtypes = [lay, i32]
    L.sfb_mpc_record_doubles.restype = i64
    L.sfb_mpc_shared_jac_doubles.argtypes = [lay]
    L.sfb_mpc_shared_jac_doubles.restype = i64
    L.sfb_mpc_nnzA.argtypes = [lay]
    L.sfb_mpc_nnzA.restype = i64
    L.sfb_mpc_assemble_batch.argtypes = [lay, i64, dp, dp, dp, dp, dp, vp]
    L.sfb_mpc_swarm_create.argtypes = [vp, lay, dp, dp, i64, C.POINTER(C.c_void_p)]
    L.sfb_mpc_swarm_destroy.argtypes = [vp]
    L.sfb_mpc_swarm_destroy.restype = None
    L.sfb_mpc_swarm_reset_warmstart.argtypes = [vp]
    L.sfb_mpc_swarm_set_jac_keep.argtypes = [vp, vp, C.POINTER(C.c_int64)]
    L.sfb_mpc_swarm_host_records.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.sfb_mpc_swarm_upload.argtypes = [vp, i64, i64]
    L.sfb_mpc_swarm_step_host.argtypes = [vp, C.POINTER(SfbQPParams), dp, dp, i32, dp, dp, dp, dp, dp]
    L.sfb_mpc_swarm_device_records.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.sfb_mpc_swarm_step_resident.argtypes = [vp, C.POINTER(SfbQPParams), i32, dp, dp, dp, dp, dp]
    L.sfb_mpc_swarm_debug_buffers.argtypes = [vp] + [C.POINTER(C.c_void_p)] * 3
    L.sfb_set_devices.argtypes = [C.POINTER(C.c_int), i32]
    L.sfb_get_devices.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(C.c_int)]
    L.sfb_host_staging_trim.restype = None
    L.sfb_random_qp_batch.argtypes = [C.c_uint32, i64, i32, i32, C.c_double] + [dp] * 5
    return L


lib = _load()


def debug_set(name, value=None):
    """sfb_debug_set: a debug knob of the library (csrc/knobs.h) -- launch shapes / engine choices for tests and A/B
    measurements; value None clears it.  The library reads no environment variable: this is the only way in."""
    check(lib.sfb_debug_set(name.encode(), None if value is None else str(value).encode()))


def debug_set_from(spec):
    """"NAME=VALUE,NAME=VALUE" (the KNOBS variable of the scripts, bench.py --debug-knob): applies every pair, returns them."""
    pairs = [kv.split("=", 1) for kv in spec.replace(";", ",").split(",") if kv.strip()]
    for k, v in pairs:
        debug_set(k.strip(), v.strip())
    return {k.strip(): v.strip() for k, v in pairs}


def check(status):
    if status != SFB_OK:
        raise SfbError(status, lib.sfb_last_error().decode())


def device_count():
    n = C.c_int(0)
    check(lib.sfb_device_count(C.byref(n)))
    return n.value


def set_devices(devices=None):
    """Device list of the *_multi entry points (None / empty: every visible device)."""
    devices = list(devices or [])
    arr = (C.c_int * max(1, len(devices)))(*devices)
    check(lib.sfb_set_devices(arr if devices else None, len(devices)))


def get_devices():
    n = C.c_int(0)
    arr = (C.c_int * 64)()
    check(lib.sfb_get_devices(arr, 64, C.byref(n)))
    return [arr[i] for i in range(min(n.value, 64))]
