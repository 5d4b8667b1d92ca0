"""Python mirror of the reference's QP interface (qp.hpp, qp_solver.hpp) over the C-ABI.

Names, argument meaning and error behaviour follow smooth::feedback:
  QuadraticProgram (qp.hpp:31-45), QPSolutionStatus (:82-92), QPSolution (:95-108),
  QPSolverParams (qp_solver.hpp:29-68), QPSolver (:242-757), solve_qp (:779-787).
The batch entry points (`solve_qp_batch`) are the data-parallel extension this engine adds.
All numerics run in the HIP kernels; nothing here computes.
"""
import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _capi


class QPSolutionStatus(enum.IntEnum):
    """qp.hpp:82-92"""
    Optimal = 0
    PolishFailed = 1
    PrimalInfeasible = 2
    DualInfeasible = 3
    MaxIterations = 4
    MaxTime = 5
    Unknown = 6


@dataclass
class QPSolverParams:
    """qp_solver.hpp:29-68 (float members are rounded to binary32 like the reference's)."""
    verbose: bool = False
    alpha: float = 1.6
    rho: float = 0.1
    sigma: float = 1e-6
    scaling: bool = True
    eps_abs: float = 1e-3
    eps_rel: float = 1e-3
    eps_primal_inf: float = 1e-4
    eps_dual_inf: float = 1e-4
    max_iter: Optional[int] = None
    max_time: Optional[float] = None  # seconds, on the device clock per item (include/sfb.h)
    stop_check_iter: int = 25
    polish: bool = True
    polish_iter: int = 5
    delta: float = 1e-6
    reuse_factor: bool = False  # extension (include/sfb.h): P and A unchanged since the previous call on the workspace

    def to_c(self):
        p = _capi.SfbQPParams()
        _capi.lib.sfb_qp_params_default(C.byref(p))
        p.verbose = int(self.verbose)
        p.reuse_factor = int(self.reuse_factor)
        p.alpha, p.rho, p.sigma = self.alpha, self.rho, self.sigma
        p.scaling = int(self.scaling)
        p.eps_abs, p.eps_rel = self.eps_abs, self.eps_rel
        p.eps_primal_inf, p.eps_dual_inf = self.eps_primal_inf, self.eps_dual_inf
        p.max_iter = -1 if self.max_iter is None else int(self.max_iter)
        p.max_time_ns = -1 if self.max_time is None else int(self.max_time * 1e9)
        p.stop_check_iter = int(self.stop_check_iter)
        p.polish = int(self.polish)
        p.polish_iter = int(self.polish_iter)
        p.delta = self.delta
        return p


@dataclass
class QuadraticProgram:
    """qp.hpp:31-45: min 1/2 x'Px + q'x  s.t.  l <= Ax <= u.  P (n,n), q (n,), A (m,n), l,u (m,)."""
    P: np.ndarray
    q: np.ndarray
    A: np.ndarray
    l: np.ndarray
    u: np.ndarray


@dataclass
class QPSolution:
    """qp.hpp:95-108"""
    code: QPSolutionStatus = QPSolutionStatus.Unknown
    iter: int = 0
    primal: np.ndarray = field(default_factory=lambda: np.zeros(0))
    dual: np.ndarray = field(default_factory=lambda: np.zeros(0))
    objective: float = 0.0


@dataclass
class QPBatchSolution:
    code: np.ndarray
    iter: np.ndarray
    primal: np.ndarray
    dual: np.ndarray
    objective: np.ndarray
    trace: Optional[np.ndarray] = None  # (B, rows, 5): ITER, OBJ, PRI_RES, DUA_RES, TIME [us] per stopping check (ITER -1: unused)
    phase_us: Optional[np.ndarray] = None  # (B, 6): scaling | matrix filling | factorization | iteration | polish | report [us]


def _f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != shape:
        raise ValueError("expected shape %s, got %s" % (shape, a.shape))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data


def pack_colmajor(M):
    """(B, rows, cols) -> (B, rows*cols) column-major buffers (Eigen default layout, qp.hpp:35-41)."""
    M = np.asarray(M, dtype=np.float64)
    return np.ascontiguousarray(np.transpose(M, (0, 2, 1)).reshape(M.shape[0], -1))


def solve_qp_batch_host(P, q, A, l, u, prm: Optional[QPSolverParams] = None, warm_x=None, warm_y=None, multi_device=False,
                        trace_rows=0, phases=False):
    """Batched solve_qp on host numpy buffers.  P (B, n*n) and A (B, m*n) are COLUMN-major flat
    buffers (see pack_colmajor), q (B,n), l,u (B,m).  Calls sfb_qp_dense_solve_batch_host, or with multi_device the
    variant that shards the batch over the device list (_capi.set_devices).  trace_rows > 0 (n + m <= 128): the
    reference's verbose table (qp_solver.hpp:490-501) as data in `.trace` (sfb_qp_dense_solve_batch_host_trace)."""
    q = np.ascontiguousarray(q, dtype=np.float64)
    l = np.ascontiguousarray(l, dtype=np.float64)
    if q.ndim != 2 or l.ndim != 2:
        raise ValueError("q and l must be (batch, n) / (batch, m)")
    B, n = q.shape
    m = l.shape[1]
    P = _f64(P, (B, n * n))
    A = _f64(A, (B, m * n))
    u = _f64(u, (B, m))
    if (warm_x is None) != (warm_y is None):
        raise ValueError("warm_x and warm_y must be given together")
    if warm_x is not None:
        warm_x = _f64(warm_x, (B, n))
        warm_y = _f64(warm_y, (B, m))
    x = np.empty((B, n))
    y = np.empty((B, m))
    obj = np.empty(B)
    it = np.empty(B, dtype=np.uint32)
    code = np.empty(B, dtype=np.int32)
    cp = (prm or QPSolverParams()).to_c()
    if trace_rows or phases:  # phases: the per-phase times of qp_solver.hpp:550-565 as data in `.phase_us` (n + m <= 128)
        if multi_device:
            raise ValueError("trace_rows / phases and multi_device exclude each other")
        trace = np.empty((B, int(trace_rows), 5)) if trace_rows else None
        ph = np.zeros((B, 6)) if phases else None
        _capi.check(_capi.lib.sfb_qp_dense_solve_batch_host_phases(
            C.byref(cp), B, n, m, _ptr(P), _ptr(q), _ptr(A), _ptr(l), _ptr(u), _ptr(warm_x), _ptr(warm_y),
            _ptr(x), _ptr(y), _ptr(obj), _ptr(it), _ptr(code), _ptr(trace), int(trace_rows), _ptr(ph)))
        return QPBatchSolution(code=code, iter=it, primal=x, dual=y, objective=obj, trace=trace, phase_us=ph)
    fn = _capi.lib.sfb_qp_dense_solve_batch_host_multi if multi_device else _capi.lib.sfb_qp_dense_solve_batch_host
    _capi.check(fn(
        C.byref(cp), B, n, m, _ptr(P), _ptr(q), _ptr(A), _ptr(l), _ptr(u), _ptr(warm_x), _ptr(warm_y),
        _ptr(x), _ptr(y), _ptr(obj), _ptr(it), _ptr(code)))
    return QPBatchSolution(code=code, iter=it, primal=x, dual=y, objective=obj)


def solve_qp_batch_device(B, n, m, dP, dq, dA, dl, du, dx, dy, dobj, diter, dcode, prm=None,
                          dwarm_x=0, dwarm_y=0, stream=0):
    """Asynchronous batched solve on DEVICE pointers (ints), e.g. torch tensors' data_ptr().
    Calls sfb_qp_dense_solve_batch on `stream` (a hipStream_t handle as int, 0 = default)."""
    cp = (prm or QPSolverParams()).to_c()
    _capi.check(_capi.lib.sfb_qp_dense_solve_batch(
        C.byref(cp), B, n, m, dP, dq, dA, dl, du, dwarm_x or None, dwarm_y or None, dx, dy,
        dobj or None, diter or None, dcode, stream or None))


class Workspace:
    """sfb_workspace: device memory the caller creates once and hands to every call (no allocation per call).
    Workspace.for_dense(batch, n, m, prm) sizes it with sfb_qp_dense_workspace_bytes."""

    def __init__(self, nbytes):
        self._h = C.c_void_p()
        _capi.check(_capi.lib.sfb_workspace_create(int(nbytes), C.byref(self._h)))
        p, b = C.c_void_p(), C.c_int64()
        _capi.check(_capi.lib.sfb_workspace_info(self._h, C.byref(p), C.byref(b)))
        self.device_ptr, self.nbytes = p.value or 0, b.value

    @staticmethod
    def dense_bytes(batch, n, m, prm=None):
        v = C.c_int64()
        cp = (prm or QPSolverParams()).to_c()
        _capi.check(_capi.lib.sfb_qp_dense_workspace_bytes(C.byref(cp), int(batch), int(n), int(m), C.byref(v)))
        return v.value

    @classmethod
    def for_dense(cls, batch, n, m, prm=None):
        return cls(cls.dense_bytes(batch, n, m, prm))

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib.sfb_workspace_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_qp_batch_device_ws(B, n, m, dP, dq, dA, dl, du, dx, dy, dobj, diter, dcode, workspace: "Workspace", prm=None,
                             dwarm_x=0, dwarm_y=0, stream=0):
    """sfb_qp_dense_solve_batch_ws: like solve_qp_batch_device on the caller's Workspace."""
    cp = (prm or QPSolverParams()).to_c()
    _capi.check(_capi.lib.sfb_qp_dense_solve_batch_ws(
        C.byref(cp), B, n, m, dP, dq, dA, dl, du, dwarm_x or None, dwarm_y or None, dx, dy,
        dobj or None, diter or None, dcode, workspace._h, stream or None))


class QPSolver:
    """QPSolver<QuadraticProgram<M,N,double>>, qp_solver.hpp:242-757 (dense problems, n+m <= 64)."""

    def __init__(self, pbm: Optional[QuadraticProgram] = None, prm: Optional[QPSolverParams] = None):
        self.prm_ = prm or QPSolverParams()
        self.sol_ = QPSolution()
        if pbm is not None:
            self.analyze(pbm)

    def analyze(self, pbm: QuadraticProgram):
        """qp_solver.hpp:297-338: size the solution (the device kernel owns its workspace in LDS)."""
        A = np.asarray(pbm.A)
        self.sol_.primal = np.zeros(A.shape[1])
        self.sol_.dual = np.zeros(A.shape[0])

    def sol(self) -> QPSolution:
        return self.sol_

    def solve(self, pbm: QuadraticProgram, warmstart: Optional[QPSolution] = None) -> QPSolution:
        """qp_solver.hpp:343-568"""
        P = np.asarray(pbm.P, dtype=np.float64)
        A = np.asarray(pbm.A, dtype=np.float64)
        m, n = A.shape
        wx = wy = None
        if warmstart is not None:
            wx = np.asarray(warmstart.primal, dtype=np.float64).reshape(1, n)
            wy = np.asarray(warmstart.dual, dtype=np.float64).reshape(1, m)
        r = solve_qp_batch_host(
            pack_colmajor(P.reshape(1, n, n)), np.asarray(pbm.q, dtype=np.float64).reshape(1, n),
            pack_colmajor(A.reshape(1, m, n)), np.asarray(pbm.l, dtype=np.float64).reshape(1, m),
            np.asarray(pbm.u, dtype=np.float64).reshape(1, m), self.prm_, wx, wy)
        self.sol_ = QPSolution(code=QPSolutionStatus(int(r.code[0])), iter=int(r.iter[0]), primal=r.primal[0],
                               dual=r.dual[0], objective=float(r.objective[0]))
        return self.sol_


def solve_qp(pbm: QuadraticProgram, prm: Optional[QPSolverParams] = None,
             warmstart: Optional[QPSolution] = None) -> QPSolution:
    """qp_solver.hpp:779-787"""
    return QPSolver(pbm, prm).solve(pbm, warmstart)


def random_qp_batch(seed, batch, m, n, density):
    """benchmarks/bench_types.hpp:19-41 drawn `batch` times from std::default_random_engine(seed).
    Returns flat col-major buffers (P, q, A, l, u) as in solve_qp_batch_host."""
    P = np.empty((batch, n * n))
    q = np.empty((batch, n))
    A = np.empty((batch, m * n))
    l = np.empty((batch, m))
    u = np.empty((batch, m))
    _capi.check(_capi.lib.sfb_random_qp_batch(seed, batch, m, n, float(density), _ptr(P), _ptr(q), _ptr(A),
                                              _ptr(l), _ptr(u)))
    return P, q, A, l, u


# ------------------------------------------------------------------------------------------------
# Sparse problems sharing one pattern: QuadraticProgramSparse (qp.hpp:60-79) and the sparse
# instantiation of QPSolver (qp_solver.hpp sparse branches), as used by MPC::operator() (mpc.hpp:491).
# ------------------------------------------------------------------------------------------------
@dataclass
class QuadraticProgramSparse:
    """qp.hpp:60-79.  P: scipy.sparse CSC (as stored; only col >= row enters the KKT matrix),
    A: scipy.sparse CSR, q (n,), l,u (m,)."""
    P: object
    q: np.ndarray
    A: object
    l: np.ndarray
    u: np.ndarray


class SparseQPPlan:
    """Symbolic analysis shared by a batch: QPSolver::analyze + SimplicialLDLT::analyzePattern
    (qp_solver.hpp:297-338, :424).  Host-only; owns an sfb_sparse_qp_plan."""

    def __init__(self, n, m, P_colptr, P_rowind, A_rowptr, A_colind, ordering=1, user_perm=None, stage=None,
                 keep=None):
        """keep (nnzA bools, optional): False = the stored entry of A is zero in every item solved with this plan
        (sfb_sparse_qp_plan_create_pruned: analysed without it, checked per item on the device, whole-pattern
        fallback for items that violate it)."""
        self.n, self.m = int(n), int(m)
        self._keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (P_colptr, P_rowind, A_rowptr, A_colind)]
        up = None if user_perm is None else np.ascontiguousarray(user_perm, dtype=np.int32)
        st = None if stage is None else np.ascontiguousarray(stage, dtype=np.int32)
        if st is not None and st.shape != (self.n + self.m,):
            raise ValueError("stage must have n+m entries")
        kp = None
        if keep is not None:
            kp = np.ascontiguousarray(np.asarray(keep) != 0, dtype=np.uint8)
            if kp.shape != (int(self._keep[2][-1]),):
                raise ValueError("keep must have nnzA entries")
        h = C.c_void_p()
        _capi.check(_capi.lib.sfb_sparse_qp_plan_create_pruned(
            self.n, self.m, *[_ptr(a) if a.size else None for a in self._keep], int(ordering), _ptr(up), _ptr(st),
            _ptr(kp), C.byref(h)))
        self._h = h
        a, b, c_ = C.c_int64(), C.c_int64(), C.c_int64()
        _capi.check(_capi.lib.sfb_sparse_qp_plan_info(h, C.byref(a), C.byref(b), C.byref(c_)))
        self.nnzK, self.nnzL, self.workspace_bytes_per_item = a.value, b.value, c_.value
        self.nnzP, self.nnzA = int(self._keep[0][-1]), int(self._keep[2][-1])
        _capi.check(_capi.lib.sfb_sparse_qp_plan_pruned_info(h, C.byref(a), C.byref(b)))
        self.nnzA_kept, self.nnzL_fallback = a.value, b.value
        self.pruned = self.nnzA_kept != self.nnzA

    def workspace_bytes(self, batch):
        """Exact device workspace of one call (sfb_sparse_qp_plan_workspace_bytes)."""
        v = C.c_int64()
        _capi.check(_capi.lib.sfb_sparse_qp_plan_workspace_bytes(self._h, int(batch), C.byref(v)))
        return v.value

    @classmethod
    def from_scipy(cls, P, A, **kw):
        import scipy.sparse as sp
        P = sp.csc_matrix(P); P.sort_indices()
        A = sp.csr_matrix(A); A.sort_indices()
        return cls(A.shape[1], A.shape[0], P.indptr, P.indices, A.indptr, A.indices, **kw)

    @property
    def perm(self):
        p = np.empty(self.n + self.m, dtype=np.int32)
        _capi.check(_capi.lib.sfb_sparse_qp_plan_get_perm(self._h, _ptr(p)))
        return p

    def factor_order(self, fallback=False):
        """Rank of every permuted column in the summation order of the numeric factorisation
        (sfb_sparse_qp_plan_get_factor_order); fallback: of a pruned plan's whole-pattern analysis."""
        r = np.empty(self.n + self.m, dtype=np.int32)
        _capi.check(_capi.lib.sfb_sparse_qp_plan_get_factor_order(self._h, int(bool(fallback)), _ptr(r)))
        return r

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib.sfb_sparse_qp_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve_batch_host(self, Px, q, Ax, l, u, prm: Optional[QPSolverParams] = None, warm_x=None, warm_y=None,
                         multi_device=False, trace_rows=0, phases=False, out: Optional["QPBatchSolution"] = None):
        """sfb_sparse_qp_solve_batch_host[_multi | _trace]: Px (B, nnzP), q (B, n), Ax (B, nnzA), l,u (B, m).
        trace_rows > 0: the reference's verbose table (qp_solver.hpp:490-501) as data in `.trace`.
        out: a QPBatchSolution of an earlier call with the same batch size whose arrays receive the results (the C entry writes
        into the caller's buffers; a control loop keeps them from tick to tick instead of touching fresh pages every call)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        Px = _f64(np.reshape(Px, (B, self.nnzP)), (B, self.nnzP))
        Ax = _f64(np.reshape(Ax, (B, self.nnzA)), (B, self.nnzA))
        q = _f64(q, (B, self.n)); l = _f64(l, (B, self.m)); u = _f64(u, (B, self.m))
        if (warm_x is None) != (warm_y is None):
            raise ValueError("warm_x and warm_y must be given together")
        if warm_x is not None:
            warm_x = _f64(warm_x, (B, self.n)); warm_y = _f64(warm_y, (B, self.m))
        if out is not None:
            x, y, obj, it, code = out.primal, out.dual, out.objective, out.iter, out.code
            for a, shp, dt in ((x, (B, self.n), np.float64), (y, (B, self.m), np.float64), (obj, (B,), np.float64),
                               (it, (B,), np.uint32), (code, (B,), np.int32)):
                if a.shape != shp or a.dtype != dt or not a.flags.c_contiguous or not a.flags.writeable:
                    raise ValueError("out: arrays of another batch size or layout")
        else:
            x = np.empty((B, self.n)); y = np.empty((B, self.m)); obj = np.empty(B)
            it = np.empty(B, dtype=np.uint32); code = np.empty(B, dtype=np.int32)
        cp = (prm or QPSolverParams()).to_c()
        if trace_rows or phases:  # phases: the per-phase times of qp_solver.hpp:550-565 as data in `.phase_us`
            if multi_device:
                raise ValueError("trace_rows / phases and multi_device exclude each other")
            trace = np.empty((B, int(trace_rows), 5)) if trace_rows else None
            ph = np.zeros((B, 6)) if phases else None
            _capi.check(_capi.lib.sfb_sparse_qp_solve_batch_host_phases(
                self._h, C.byref(cp), B, _ptr(Px), _ptr(q), _ptr(Ax), _ptr(l), _ptr(u), _ptr(warm_x), _ptr(warm_y),
                _ptr(x), _ptr(y), _ptr(obj), _ptr(it), _ptr(code), _ptr(trace), int(trace_rows), _ptr(ph)))
            return QPBatchSolution(code=code, iter=it, primal=x, dual=y, objective=obj, trace=trace, phase_us=ph)
        fn = _capi.lib.sfb_sparse_qp_solve_batch_host_multi if multi_device else _capi.lib.sfb_sparse_qp_solve_batch_host
        _capi.check(fn(
            self._h, C.byref(cp), B, _ptr(Px), _ptr(q), _ptr(Ax), _ptr(l), _ptr(u), _ptr(warm_x), _ptr(warm_y),
            _ptr(x), _ptr(y), _ptr(obj), _ptr(it), _ptr(code)))
        return QPBatchSolution(code=code, iter=it, primal=x, dual=y, objective=obj)

    def solve_batch_device(self, B, dPx, dq, dAx, dl, du, dx, dy, dobj, diter, dcode, dworkspace, prm=None,
                           dwarm_x=0, dwarm_y=0, stream=0, dorder=0):
        """sfb_sparse_qp_solve_batch[_ordered] on device pointers (ints); asynchronous on `stream`.
        dorder: int32 permutation (launch position -> item), 0 = natural order."""
        cp = (prm or QPSolverParams()).to_c()
        _capi.check(_capi.lib.sfb_sparse_qp_solve_batch_ordered(
            self._h, C.byref(cp), B, dPx, dq, dAx, dl, du, dwarm_x or None, dwarm_y or None, dx, dy,
            dobj or None, diter or None, dcode, dworkspace, dorder or None, stream or None))


    def solve_batch_device_phases(self, B, dPx, dq, dAx, dl, du, dx, dy, dobj, diter, dcode, dworkspace, dphase_us, prm=None,
                                  dwarm_x=0, dwarm_y=0, stream=0, dtrace=0, trace_rows=0):
        """sfb_sparse_qp_solve_batch_phases on device pointers: the same solve through the TRACE instance of the kernel (one
        wave per item), with the per-phase microseconds of qp_solver.hpp:550-565 in dphase_us (B x 6 doubles)."""
        cp = (prm or QPSolverParams()).to_c()
        _capi.check(_capi.lib.sfb_sparse_qp_solve_batch_phases(
            self._h, C.byref(cp), B, dPx, dq, dAx, dl, du, dwarm_x or None, dwarm_y or None, dx, dy,
            dobj or None, diter or None, dcode, dworkspace, dtrace or None, int(trace_rows), dphase_us, stream or None))


def solve_qp_sparse(pbm: QuadraticProgramSparse, prm: Optional[QPSolverParams] = None,
                    warmstart: Optional[QPSolution] = None) -> QPSolution:
    """solve_qp for QuadraticProgramSparse (qp_solver.hpp:779-787, sparse instantiation)."""
    import scipy.sparse as sp
    P = sp.csc_matrix(pbm.P); P.sort_indices()
    A = sp.csr_matrix(pbm.A); A.sort_indices()
    plan = SparseQPPlan(A.shape[1], A.shape[0], P.indptr, P.indices, A.indptr, A.indices)
    wx = wy = None
    if warmstart is not None:
        wx = np.asarray(warmstart.primal, dtype=np.float64)[None]
        wy = np.asarray(warmstart.dual, dtype=np.float64)[None]
    r = plan.solve_batch_host(P.data[None], np.asarray(pbm.q, float)[None], A.data[None],
                              np.asarray(pbm.l, float)[None], np.asarray(pbm.u, float)[None], prm, wx, wy)
    plan.close()
    return QPSolution(code=QPSolutionStatus(int(r.code[0])), iter=int(r.iter[0]), primal=r.primal[0], dual=r.dual[0],
                      objective=float(r.objective[0]))
