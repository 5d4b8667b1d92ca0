"""Batched EKF covariance steps over the C-ABI (matrix part of smooth::feedback::EKF, ekf.hpp:79-139).

The Lie-group side (A = -ad(f) + d^r f/dx, H = d^r h/dx, r = y (-) h(g), g <- g (+) delta) is the
caller's; see include/smooth_feedback_amd/ekf.hpp for the C++ front that mirrors EKF<G>."""
import numpy as np

from . import _capi
from .qp import _ptr


def _mat(a, B, w, shared_ok=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shared_ok and a.size == w:
        return a.reshape(w), 1
    if a.shape != (B, w):
        a = a.reshape(B, w)
    return a, 0


def ekf_step_batch_host(P, dof, A=None, Q=None, dt=None, H=None, R=None, r=None):
    """One fused step on host buffers (sfb_ekf_step_batch_host).  P (B, dof*dof) col-major flat.
    predict part if A is given: A (B, dof*dof), Q (B, dof*dof) or (dof*dof,) shared, dt (B,) or scalar.
    update part if H is given: H (B, ny*dof), R (B, ny*ny) or shared, r (B, ny).
    Returns (P_new, delta or None, info or None)."""
    P = np.array(P, dtype=np.float64, order="C")
    B = P.shape[0]
    nn = dof * dof
    assert P.shape == (B, nn)
    pa = pq = pdt = ph = pr = prr = None
    qs = ds = rs = 0
    ny = 1
    keep = []
    if A is not None:
        A, _ = _mat(A, B, nn); Q, qs = _mat(Q, B, nn, True)
        dt = np.ascontiguousarray(np.atleast_1d(dt), dtype=np.float64)
        ds = int(dt.size == 1 and B != 1) or int(dt.size == 1)
        keep += [A, Q, dt]
        pa, pq, pdt = _ptr(A), _ptr(Q), _ptr(dt)
    delta = info = None
    if H is not None:
        r = np.ascontiguousarray(r, dtype=np.float64)
        ny = r.shape[1]
        H, _ = _mat(H, B, ny * dof); R, rs = _mat(R, B, ny * ny, True)
        delta = np.zeros((B, dof)); info = np.zeros(B, dtype=np.int32)
        keep += [H, R, r]
        ph, pr, prr = _ptr(H), _ptr(R), _ptr(r)
    _capi.check(_capi.lib.sfb_ekf_step_batch_host(B, dof, ny, pa, pq, qs, pdt, ds, ph, pr, rs, prr, _ptr(P),
                                                  _ptr(delta) if delta is not None else None,
                                                  _ptr(info) if info is not None else None))
    return P, delta, info


STEPPERS = {"euler": 0, "rk4": 1}  # sfb_ekf_stepper


def ekf_predict_batch_host(P, dof, A, Q, dt, stepper="euler", A_mid=None, A_end=None):
    """One predict step on host buffers with the chosen stepper (sfb_ekf_predict_stepper_batch_host):
    "euler" (ekf.hpp:30 default) or "rk4" (odeint runge_kutta4, tests/test_ekf.cpp:113-115).  A_mid / A_end (rk4):
    the linearisation at t + dt/2 and t + dt for dynamics that depend on t (sfb_ekf_predict_rk4_batch_host).
    Returns P_new."""
    P = np.array(P, dtype=np.float64, order="C")
    B, nn = P.shape[0], dof * dof
    A, _ = _mat(A, B, nn); Q, qs = _mat(Q, B, nn, True)
    dt = np.ascontiguousarray(np.atleast_1d(dt), dtype=np.float64)
    if A_mid is not None:
        if stepper != "rk4":
            raise ValueError("A_mid / A_end are runge_kutta4 stage matrices")
        A_mid, _ = _mat(A_mid, B, nn); A_end, _ = _mat(A_end, B, nn)
        _capi.check(_capi.lib.sfb_ekf_predict_rk4_batch_host(B, dof, _ptr(A), _ptr(A_mid), _ptr(A_end), _ptr(Q), qs, _ptr(dt),
                                                             int(dt.size == 1), _ptr(P)))
        return P
    _capi.check(_capi.lib.sfb_ekf_predict_stepper_batch_host(STEPPERS[stepper], B, dof, _ptr(A), _ptr(Q), qs, _ptr(dt),
                                                             int(dt.size == 1), _ptr(P)))
    return P


def ekf_predict_stepper_batch_device(stepper, B, dof, dA, dQ, q_shared, ddt, dt_shared, dP, stream=0):
    _capi.check(_capi.lib.sfb_ekf_predict_stepper_batch(STEPPERS[stepper], B, dof, dA, dQ, int(q_shared), ddt,
                                                        int(dt_shared), dP, stream or None))


def ekf_predict_update_batch_device(B, dof, ny, dA, dQ, q_shared, ddt, dt_shared, dH, dR, r_shared, dr, dP, ddelta,
                                    dinfo=0, stream=0):
    """sfb_ekf_predict_update_batch on device pointers (ints), asynchronous on `stream`."""
    _capi.check(_capi.lib.sfb_ekf_predict_update_batch(B, dof, ny, dA, dQ, int(q_shared), ddt, int(dt_shared), dH, dR,
                                                       int(r_shared), dr, dP, ddelta, dinfo or None, stream or None))


def ekf_predict_batch_device(B, dof, dA, dQ, q_shared, ddt, dt_shared, dP, stream=0):
    _capi.check(_capi.lib.sfb_ekf_predict_batch(B, dof, dA, dQ, int(q_shared), ddt, int(dt_shared), dP, stream or None))


def ekf_update_batch_device(B, dof, ny, dH, dR, r_shared, dr, dP, ddelta, dinfo=0, stream=0):
    _capi.check(_capi.lib.sfb_ekf_update_batch(B, dof, ny, dH, dR, int(r_shared), dr, dP, ddelta, dinfo or None,
                                               stream or None))
