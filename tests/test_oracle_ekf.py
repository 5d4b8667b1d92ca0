"""Pins the EKF CPU oracle (oracle/ekf_oracle.c) against the reference's own analytic checks:
tests/test_ekf.cpp:50-103 (UpdateLinear: textbook linear Kalman update) and :105-153 (PredictLinear:
covariance propagation vs expm(A tau); here with the default Euler stepper and a small step)."""
import numpy as np
import pytest
import scipy.linalg as sl


@pytest.mark.parametrize("Nx,Ny", [(3, 3), (10, 3), (3, 10), (6, 3)])
def test_update_linear(oracle, Nx, Ny):
    rng = np.random.default_rng(Nx * 10 + Ny)
    for _ in range(10):
        x, xhat = rng.uniform(-1, 1, Nx), rng.uniform(-1, 1, Nx)
        P = np.diag(rng.uniform(-1, 1, Nx) + 1.1)
        H, h = rng.uniform(-1, 1, (Ny, Nx)), rng.uniform(-1, 1, Ny)
        R = np.diag(rng.uniform(-1, 1, Ny) + 1.1)
        r = (H @ x + h) - (H @ xhat + h)
        Pn, delta, info = oracle.ekf_update_batch(H.flatten("F")[None], R.flatten("F")[None], r[None],
                                                  P.flatten("F")[None], Nx)
        S = H @ P @ H.T + R
        K = P @ H.T @ np.linalg.inv(S)
        assert info[0] == 0
        assert np.allclose(xhat + delta[0], xhat + K @ r, rtol=1e-6, atol=1e-12)          # :100
        assert np.allclose(Pn[0].reshape(Nx, Nx, order="F"), (np.eye(Nx) - K @ H) @ P, rtol=1e-6, atol=1e-12)  # :101


@pytest.mark.parametrize("Nx", [3, 6])
def test_predict_linear(oracle, Nx):
    rng = np.random.default_rng(Nx)
    A = rng.uniform(-1, 1, (Nx, Nx))
    P = np.diag(rng.uniform(-1, 1, Nx) + 1.1)
    tau, steps = 0.7, 7000
    Pc = P.flatten("F")[None].copy()
    for _ in range(steps):
        Pc = oracle.ekf_predict_batch(A.flatten("F")[None], np.zeros(Nx * Nx), tau / steps, Pc)
    F = sl.expm(A * tau)
    assert np.allclose(Pc[0].reshape(Nx, Nx, order="F"), F @ P @ F.T, rtol=1e-3, atol=1e-3)     # :149 tolerance


def test_predict_uses_upper_triangle_of_the_sum(oracle):
    """ekf.hpp:88: dcov = (A cov + cov A' + Q).selfadjointView<Upper>(): a non-symmetric Q only
    contributes through its upper triangle."""
    rng = np.random.default_rng(5)
    n = 4
    A, P, Q = rng.uniform(-1, 1, (n, n)), np.eye(n), rng.uniform(-1, 1, (n, n))
    Pn = oracle.ekf_predict_batch(A.flatten("F")[None], Q.flatten("F")[None], 0.1, P.flatten("F")[None])
    S = A @ P + P @ A.T + Q
    S = np.triu(S) + np.triu(S, 1).T
    assert np.allclose(Pn[0].reshape(n, n, order="F"), P + 0.1 * S, atol=1e-15)


@pytest.mark.parametrize("Nx", [3, 6, 9])
def test_predict_linear_rk4(oracle, Nx):
    """tests/test_ekf.cpp:104-153 as written there: runge_kutta4 stepper, dt = 1e-3, tau = 0.7, Q = 0 ->
    P(tau) = F P F', F = expm(A tau).  RK4 at this step reaches the exact solution to ~1e-12; the reference
    asserts 1e-3."""
    rng = np.random.default_rng(40 + Nx)
    A = rng.uniform(-1, 1, (Nx, Nx))
    P = np.diag(rng.uniform(-1, 1, Nx) + 1.1)
    tau, dt = 0.7, 1e-3
    Pc = P.flatten("F")[None].copy()
    t = 0.0
    while t + dt < tau:                      # ekf.hpp:93-101: fixed steps, then the remainder
        Pc = oracle.ekf_predict_batch(A.flatten("F")[None], np.zeros(Nx * Nx), dt, Pc, stepper="rk4")
        t += dt
    Pc = oracle.ekf_predict_batch(A.flatten("F")[None], np.zeros(Nx * Nx), tau - t, Pc, stepper="rk4")
    F = sl.expm(A * tau)
    assert np.allclose(Pc[0].reshape(Nx, Nx, order="F"), F @ P @ F.T, rtol=1e-9, atol=1e-9)


def test_rk4_step_matches_a_numpy_transcription(oracle):
    """One step against the textbook RK4 formulas in numpy (symmetrised right-hand side, Q != 0)."""
    rng = np.random.default_rng(3)
    n, dt = 6, 0.05
    A, Q = rng.uniform(-1, 1, (n, n)), np.diag(rng.uniform(0.1, 1.0, n))
    G = rng.uniform(-1, 1, (n, n)); P = np.eye(n) + G @ G.T / n

    def f(Pm):
        S = A @ Pm + Pm @ A.T + Q
        return np.triu(S) + np.triu(S, 1).T
    k1 = f(P); k2 = f(P + dt / 2 * k1); k3 = f(P + dt / 2 * k2); k4 = f(P + dt * k3)
    ref = P + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    got = oracle.ekf_predict_batch(A.flatten("F")[None], Q.flatten("F")[None], dt, P.flatten("F")[None], stepper="rk4")
    assert np.allclose(got[0].reshape(n, n, order="F"), ref, rtol=1e-13, atol=1e-13)


def test_rk4_with_stage_linearisations_oracle(oracle):
    """oracle_ekf_predict_rk4_tv: A at the stage times (cov_ode re-linearises f(t_stage, .), ekf.hpp:84-89).  For
    A(t) = a(t) A0 the exact propagation is expm(s A0) P expm(s A0)' with s = int a dt; the stage-wise step is
    fourth-order accurate, the frozen-A step only first-order in the variation of a."""
    import scipy.linalg as sl
    rng = np.random.default_rng(8)
    n, t0 = 4, 0.2
    A0 = rng.uniform(-1, 1, (n, n)); G = rng.uniform(-1, 1, (n, n)); P0 = np.eye(n) + G @ G.T / n
    a = lambda t: 1.0 + 3.0 * t
    errs = []
    for h in (0.1, 0.05):
        fl = lambda M: np.ascontiguousarray(M.flatten("F"))[None]
        got = oracle.ekf_predict_batch(fl(a(t0) * A0), np.zeros(n * n), h, fl(P0), stepper="rk4",
                                       A_mid=fl(a(t0 + h / 2) * A0), A_end=fl(a(t0 + h) * A0))
        s_int = h * (1.0 + 3.0 * (t0 + h / 2))
        exact = sl.expm(s_int * A0) @ P0 @ sl.expm(s_int * A0).T
        errs.append(np.abs(got.reshape(n, n).T - exact).max())
    assert errs[0] < 1e-3 and errs[1] < errs[0] / 12       # ~ h^5 per step
    same = oracle.ekf_predict_batch(fl(A0), np.zeros(n * n), 0.05, fl(P0), stepper="rk4", A_mid=fl(A0), A_end=fl(A0))
    assert np.array_equal(same, oracle.ekf_predict_batch(fl(A0), np.zeros(n * n), 0.05, fl(P0), stepper="rk4"))
