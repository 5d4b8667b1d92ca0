"""ctypes loader for the CPU oracle (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (smooth_feedback_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "liboracle.so")
# Variants of the same sources (oracle/Makefile), for tests/test_parity_robustness.py only:
#   "strict" (default, what everything else uses): -ffp-contract=off, fma only where spelled -- the arithmetic the HIP
#   kernels reproduce bit for bit;  "fast": the reference's own benchmark flags;  "san": ASan + UBSan.
# SFB_ORACLE_VARIANT selects the variant a fresh process starts with (the sanitizer build needs libasan preloaded).
_VARIANT_LIBS = {"strict": ("liboracle.so", None), "fast": ("liboracle_fast.so", "fast"), "san": ("liboracle_san.so", "san")}
_variant = os.environ.get("SFB_ORACLE_VARIANT", "strict")


class OracleQPParams(C.Structure):
    """oracle_qp_params (oracle/qp_oracle.h) == QPSolverParams, qp_solver.hpp:29-68."""

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


def build(force=False, variant="strict"):
    """Compile the oracle with the committed Makefile (gcc only)."""
    name, target = _VARIANT_LIBS[variant]
    path = os.path.join(_HERE, "build", name)
    if force or not os.path.exists(path):
        if force and os.path.exists(path) and target is not None:
            os.remove(path)
        subprocess.check_call(["make", "-C", _HERE, "-s"] + ([target] if target else []))
    return path


_libs = {}


class variant:
    """`with loader.variant("fast"): ...` -- every call inside goes to that build of the oracle."""

    def __init__(self, name):
        assert name in _VARIANT_LIBS, name
        self.name = name

    def __enter__(self):
        global _variant
        self.prev, _variant = _variant, self.name
        return self

    def __exit__(self, *exc):
        global _variant
        _variant = self.prev
        return False


def lib():
    if _variant not in _libs:
        L = C.CDLL(build(variant=_variant))
        dp = C.POINTER(C.c_double)
        L.oracle_qp_params_default.argtypes = [C.POINTER(OracleQPParams)]
        L.oracle_qp_params_default.restype = None
        L.oracle_qp_dense_solve_batch.argtypes = [
            C.POINTER(OracleQPParams), C.c_int64, C.c_int, C.c_int,
            dp, dp, dp, dp, dp, dp, dp, dp, dp, dp,
            C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_int,
        ]
        L.oracle_qp_dense_solve_batch.restype = C.c_int
        ip = C.POINTER(C.c_int32)
        L.oracle_qp_sparse_solve_batch.argtypes = [
            C.POINTER(OracleQPParams), C.c_int64, C.c_int, C.c_int, ip, ip, dp, dp, ip, ip, dp, dp, dp,
            ip, dp, dp, dp, dp, dp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_int,
            C.POINTER(C.c_int64)]
        L.oracle_qp_sparse_solve_batch.restype = C.c_int
        L.oracle_qp_sparse_solve_batch_ordered.argtypes = [
            C.POINTER(OracleQPParams), C.c_int64, C.c_int, C.c_int, ip, ip, dp, dp, ip, ip, dp, dp, dp,
            ip, ip, dp, dp, dp, dp, dp, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_int,
            C.POINTER(C.c_int64)]
        L.oracle_qp_sparse_solve_batch_ordered.restype = C.c_int
        L.oracle_ekf_predict_batch.argtypes = [C.c_int64, C.c_int, dp, dp, C.c_int, dp, C.c_int, dp]
        L.oracle_ekf_predict_batch.restype = None
        L.oracle_ekf_predict_rk4_batch.argtypes = [C.c_int64, C.c_int, dp, dp, C.c_int, dp, C.c_int, dp]
        L.oracle_ekf_predict_rk4_batch.restype = None
        L.oracle_ekf_predict_rk4_tv_batch.argtypes = [C.c_int64, C.c_int, dp, dp, dp, dp, C.c_int, dp, C.c_int, dp]
        L.oracle_ekf_predict_rk4_tv_batch.restype = None
        L.oracle_ekf_update_batch.argtypes = [C.c_int64, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, dp,
                                              C.POINTER(C.c_int32)]
        L.oracle_ekf_update_batch.restype = None
        L.oracle_ldlt_factor.argtypes = [C.c_int, dp, C.c_int, C.POINTER(C.c_int)]
        L.oracle_ldlt_factor.restype = C.c_int
        L.oracle_ldlt_solve.argtypes = [C.c_int, dp, C.c_int, C.POINTER(C.c_int), dp]
        L.oracle_ldlt_solve.restype = None
        _libs[_variant] = L
    return _libs[_variant]


def default_params(**kw):
    p = OracleQPParams()
    lib().oracle_qp_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def qp_dense_solve_batch(P, q, A, l, u, params=None, warm_x=None, warm_y=None, nthreads=1, trace_rows=0):
    """Batch-major inputs: P (B,n,n) with P[b] stored COLUMN-major, i.e. P[b].ravel() is the
    col-major buffer (pass np.asfortranarray-style data flattened); same for A (B, m*n).

    To keep things unambiguous this function takes flat buffers: P (B, n*n), q (B, n),
    A (B, m*n), l (B, m), u (B, m), all float64 C-contiguous.
    Returns dict(x, y, obj, iter, code).
    """
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64)
    l = np.ascontiguousarray(l, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    B, n = q.shape
    m = l.shape[1]
    assert P.shape == (B, n * n) and A.shape == (B, m * n) and u.shape == (B, m)
    if warm_x is not None:
        warm_x = np.ascontiguousarray(warm_x, dtype=np.float64)
        warm_y = np.ascontiguousarray(warm_y, dtype=np.float64)
        assert warm_x.shape == (B, n) and warm_y.shape == (B, m)
    x = np.zeros((B, n))
    y = np.zeros((B, m))
    obj = np.zeros(B)
    it = np.zeros(B, dtype=np.uint32)
    code = np.zeros(B, dtype=np.int32)
    p = params if params is not None else default_params()
    trace = None
    if trace_rows:  # the verbose table of qp_solver.hpp:490-501 as data: (ITER, OBJ, PRI_RES, DUA_RES, PRI_TOL, DUA_TOL) per check
        trace = np.full((B, int(trace_rows), 6), -1.0)
        lib().oracle_qp_dense_set_trace.argtypes = [C.POINTER(C.c_double), C.c_int]
        lib().oracle_qp_dense_set_trace.restype = None
        lib().oracle_qp_dense_set_trace(_dp(trace), int(trace_rows))
    try:
        rc = lib().oracle_qp_dense_solve_batch(
            C.byref(p), B, n, m, _dp(P), _dp(q), _dp(A), _dp(l), _dp(u), _dp(warm_x), _dp(warm_y),
            _dp(x), _dp(y), _dp(obj), it.ctypes.data_as(C.POINTER(C.c_uint32)),
            code.ctypes.data_as(C.POINTER(C.c_int32)), int(nthreads))
    finally:
        if trace_rows:
            lib().oracle_qp_dense_set_trace(None, 0)
    if rc != 0:
        raise RuntimeError("oracle_qp_dense_solve_batch failed rc=%d" % rc)
    return dict(x=x, y=y, obj=obj, iter=it, code=code, trace=trace)


def colmajor(M):
    """Flatten a 2-D (rows, cols) numpy matrix into the col-major buffer the C side expects."""
    return np.asarray(M, dtype=np.float64).flatten(order="F")


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=None, params=None, warm_x=None, warm_y=None,
                          nthreads=1, forder=None, trace_rows=0):
    """Sparse branch.  P CSC (Pp, Pi) with values Px (B, nnzP); A CSR (Ap, Aj) with values Ax (B, nnzA).
    perm: elimination order of the (n+m) KKT unknowns (new -> old) or None (natural).
    forder: accumulation order of the numeric factorisation (rank of every permuted column) or None (postorder of
    the elimination tree)."""
    Pp = np.ascontiguousarray(Pp, dtype=np.int32); Pi = np.ascontiguousarray(Pi, dtype=np.int32)
    Ap = np.ascontiguousarray(Ap, dtype=np.int32); Aj = np.ascontiguousarray(Aj, dtype=np.int32)
    q = np.ascontiguousarray(q, dtype=np.float64); l = np.ascontiguousarray(l, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    B, n = q.shape
    m = l.shape[1]
    Px = np.ascontiguousarray(Px, dtype=np.float64).reshape(B, -1)
    Ax = np.ascontiguousarray(Ax, dtype=np.float64).reshape(B, -1)
    assert Px.shape[1] == Pp[n] and Ax.shape[1] == Ap[m] and len(Pp) == n + 1 and len(Ap) == m + 1
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        assert sorted(perm.tolist()) == list(range(n + m))
    if warm_x is not None:
        warm_x = np.ascontiguousarray(warm_x, dtype=np.float64)
        warm_y = np.ascontiguousarray(warm_y, dtype=np.float64)
    x = np.zeros((B, n)); y = np.zeros((B, m)); obj = np.zeros(B)
    it = np.zeros(B, dtype=np.uint32); code = np.zeros(B, dtype=np.int32)
    nnzL = C.c_int64(0)
    p = params if params is not None else default_params()
    if forder is not None:
        forder = np.ascontiguousarray(forder, dtype=np.int32)
        assert sorted(forder.tolist()) == list(range(n + m))
    trace = None
    if trace_rows:  # the verbose table of qp_solver.hpp:490-501 as data: (ITER, OBJ, PRI_RES, DUA_RES, PRI_TOL, DUA_TOL) per check
        trace = np.full((B, int(trace_rows), 6), -1.0)
        lib().oracle_qp_sparse_set_trace.argtypes = [C.POINTER(C.c_double), C.c_int]
        lib().oracle_qp_sparse_set_trace.restype = None
        lib().oracle_qp_sparse_set_trace(_dp(trace), int(trace_rows))
    rc = lib().oracle_qp_sparse_solve_batch_ordered(
        C.byref(p), B, n, m, _ip(Pp), _ip(Pi), _dp(Px), _dp(q), _ip(Ap), _ip(Aj), _dp(Ax), _dp(l), _dp(u),
        _ip(perm), _ip(forder), _dp(warm_x), _dp(warm_y), _dp(x), _dp(y), _dp(obj),
        it.ctypes.data_as(C.POINTER(C.c_uint32)), code.ctypes.data_as(C.POINTER(C.c_int32)), int(nthreads),
        C.byref(nnzL))
    if trace_rows:
        lib().oracle_qp_sparse_set_trace(None, 0)
    if rc != 0:
        raise RuntimeError("oracle_qp_sparse_solve_batch failed rc=%d" % rc)
    return dict(x=x, y=y, obj=obj, iter=it, code=code, nnzL=nnzL.value, trace=trace)


def ekf_predict_batch(A, Q, dt, P, stepper="euler", A_mid=None, A_end=None):
    """A, P: (B, dof*dof) col-major flat; Q: (B, dof*dof) or (dof*dof,) shared; dt: (B,) or scalar.
    stepper: "euler" (ekf.hpp:30 default) or "rk4" (odeint runge_kutta4).  A_mid / A_end (rk4 only): the linearisation
    at t + dt/2 and t + dt.  Returns the new P (B, dof*dof)."""
    A = np.ascontiguousarray(A, dtype=np.float64); P = np.array(P, dtype=np.float64, order="C")
    B, nn = A.shape
    dof = int(round(nn ** 0.5))
    Q = np.ascontiguousarray(Q, dtype=np.float64); dt = np.ascontiguousarray(np.atleast_1d(dt), dtype=np.float64)
    if A_mid is not None:
        assert stepper == "rk4"
        A_mid = np.ascontiguousarray(A_mid, dtype=np.float64); A_end = np.ascontiguousarray(A_end, dtype=np.float64)
        lib().oracle_ekf_predict_rk4_tv_batch(B, dof, _dp(A), _dp(A_mid), _dp(A_end), _dp(Q), int(Q.ndim == 1), _dp(dt),
                                              int(dt.size == 1), _dp(P))
        return P
    fn = lib().oracle_ekf_predict_rk4_batch if stepper == "rk4" else lib().oracle_ekf_predict_batch
    fn(B, dof, _dp(A), _dp(Q), int(Q.ndim == 1), _dp(dt), int(dt.size == 1), _dp(P))
    return P


def ekf_update_batch(H, R, r, P, dof):
    """H: (B, ny*dof), R: (B, ny*ny) or (ny*ny,) shared, r: (B, ny), P: (B, dof*dof). Returns (P_new, delta, info)."""
    H = np.ascontiguousarray(H, dtype=np.float64); r = np.ascontiguousarray(r, dtype=np.float64)
    P = np.array(P, dtype=np.float64, order="C"); R = np.ascontiguousarray(R, dtype=np.float64)
    B, ny = r.shape
    delta = np.zeros((B, dof)); info = np.zeros(B, dtype=np.int32)
    lib().oracle_ekf_update_batch(B, dof, ny, _dp(H), _dp(R), int(R.ndim == 1), _dp(r), _dp(P), _dp(delta),
                                  info.ctypes.data_as(C.POINTER(C.c_int32)))
    return P, delta, info
