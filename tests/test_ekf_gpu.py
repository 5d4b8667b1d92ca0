"""Parity of the HIP EKF kernels with the CPU oracle (bit-exact is expected: same operation order),
the reference's linear-KF identities through the device path, and size-independent properties at
BASELINE configs[4] scale.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spd(rng, B, n):
    G = rng.uniform(-1, 1, (B, n, n))
    return np.eye(n)[None] + G @ G.transpose(0, 2, 1) / n


def _flat(M):  # (B, r, c) -> col-major flat (B, r*c)
    return np.ascontiguousarray(M.transpose(0, 2, 1).reshape(M.shape[0], -1))


@pytest.mark.parametrize("dof,ny", [(6, 3), (3, 3), (2, 1), (4, 2), (6, 1), (3, 2), (2, 3)])
@pytest.mark.parametrize("B", [1, 63, 1000])
def test_predict_update_matches_oracle(sfb, oracle, dof, ny, B):
    rng = np.random.default_rng(dof * 100 + ny * 10 + B)
    P = _flat(_spd(rng, B, dof)); A = _flat(rng.uniform(-1, 1, (B, dof, dof)))
    Q = _flat(0.1 * np.tile(np.eye(dof), (B, 1, 1)) + 0.01 * rng.uniform(-1, 1, (B, dof, dof)))
    dt = rng.uniform(0.01, 0.05, B)
    H = _flat(rng.uniform(-1, 1, (B, ny, dof))); R = _flat(0.1 * np.tile(np.eye(ny), (B, 1, 1)) + 0.01 * _spd(rng, B, ny))
    r = rng.uniform(-1, 1, (B, ny))
    # separate calls
    P1, _, _ = sfb.ekf_step_batch_host(P, dof, A=A, Q=Q, dt=dt)
    ref1 = oracle.ekf_predict_batch(A, Q, dt, P)
    assert np.array_equal(P1, ref1)
    P2, d2, i2 = sfb.ekf_step_batch_host(P1, dof, H=H, R=R, r=r)
    ref2, dref, iref = oracle.ekf_update_batch(H, R, r, ref1, dof)
    assert np.array_equal(i2, iref) and np.array_equal(P2, ref2) and np.array_equal(d2, dref)
    # fused launch, shared Q / R / dt
    Qs, Rs = Q[0].copy(), R[0].copy()
    P3, d3, i3 = sfb.ekf_step_batch_host(P, dof, A=A, Q=Qs, dt=0.025, H=H, R=Rs, r=r)
    refp = oracle.ekf_predict_batch(A, Qs, 0.025, P)
    ref3, dref3, _ = oracle.ekf_update_batch(H, Rs, r, refp, dof)
    assert np.array_equal(P3, ref3) and np.array_equal(d3, dref3) and (i3 == 0).all()


@pytest.mark.parametrize("ny", [1, 2, 3])
@pytest.mark.parametrize("shared", [False, True])
def test_persistent_fused_step_matches_oracle_and_the_one_tile_kernel(sfb, oracle, knobs, ny, shared):
    """The persistent form of the fused step at dof 6 (csrc/ekf.hip ekf_fused_persistent_kernel: waves walk over tiles, the next
    tile's covariances are requested straight into LDS through a source-side swizzle) takes batches of at least four rounds of
    the device's waves; here 262 144 + 37 filters -- a partial last tile -- with per-filter and with shared Q / R / dt: against
    the oracle and against the one-tile-per-wave kernel (debug knob), bit for bit."""
    dof, B = 6, 4 * 1024 * 64 + 37
    rng = np.random.default_rng(600 + ny + 10 * shared)
    P = _flat(_spd(rng, B, dof)); A = _flat(rng.uniform(-1, 1, (B, dof, dof)))
    Q = _flat(0.1 * np.tile(np.eye(dof), (B, 1, 1)) + 0.01 * rng.uniform(-1, 1, (B, dof, dof)))
    dt = rng.uniform(0.01, 0.05, B)
    H = _flat(rng.uniform(-1, 1, (B, ny, dof))); R = _flat(0.1 * np.tile(np.eye(ny), (B, 1, 1)) + 0.01 * _spd(rng, B, ny))
    r = rng.uniform(-1, 1, (B, ny))
    Qa, Ra, dta = (Q[0].copy(), R[0].copy(), 0.025) if shared else (Q, R, dt)
    P1, d1, i1 = sfb.ekf_step_batch_host(P, dof, A=A, Q=Qa, dt=dta, H=H, R=Ra, r=r)
    knobs.set(SFB_EKF_PERSISTENT=0)
    P0, d0, i0 = sfb.ekf_step_batch_host(P, dof, A=A, Q=Qa, dt=dta, H=H, R=Ra, r=r)
    assert np.array_equal(P1, P0) and np.array_equal(d1, d0) and np.array_equal(i1, i0)
    ref, dref, iref = oracle.ekf_update_batch(H, Ra, r, oracle.ekf_predict_batch(A, Qa, dta, P), dof)
    assert np.array_equal(P1, ref) and np.array_equal(d1, dref) and np.array_equal(i1, iref)


def test_update_linear_identities_on_device(sfb):
    """tests/test_ekf.cpp:50-103 through the device path (tolerance 1e-6 as in the reference)."""
    rng = np.random.default_rng(3)
    B, Nx, Ny = 200, 3, 3
    P = np.stack([np.diag(rng.uniform(-1, 1, Nx) + 1.1) for _ in range(B)])
    H = rng.uniform(-1, 1, (B, Ny, Nx)); R = np.stack([np.diag(rng.uniform(-1, 1, Ny) + 1.1) for _ in range(B)])
    r = rng.uniform(-1, 1, (B, Ny))
    Pn, delta, info = sfb.ekf_step_batch_host(_flat(P), Nx, H=_flat(H), R=_flat(R), r=r)
    S = H @ P @ H.transpose(0, 2, 1) + R
    K = P @ H.transpose(0, 2, 1) @ np.linalg.inv(S)
    assert np.allclose(delta, np.einsum("bij,bj->bi", K, r), rtol=1e-6, atol=1e-12)
    Pexp = (np.eye(Nx)[None] - K @ H) @ P
    assert np.allclose(Pn.reshape(B, Nx, Nx).transpose(0, 2, 1), Pexp, rtol=1e-6, atol=1e-12)


def test_unsupported_sizes_fail_loudly(sfb):
    with pytest.raises(sfb._capi.SfbError) as e:
        sfb.ekf_step_batch_host(np.zeros((4, 17 * 17)), 17, H=np.zeros((4, 51)), R=np.zeros((4, 9)), r=np.zeros((4, 3)))
    assert e.value.status == sfb._capi.SFB_ERR_UNSUPPORTED


@pytest.mark.parametrize("dof,ny", [(10, 3), (3, 10), (9, 1), (5, 5), (1, 1), (16, 16), (7, 2), (6, 4), (8, 3), (9, 3), (10, 1), (10, 2),
                                    (7, 1), (9, 4), (11, 3), (6, 6), (4, 4), (6, 5)])
@pytest.mark.parametrize("B", [1, 130])
def test_generic_sizes_match_oracle(sfb, oracle, dof, ny, B):
    """The sizes outside the register-resident kernels run one filter per wavefront with the matrices in LDS --
    among them the ones the reference's own tests instantiate: test_update_linear<10,3>, <3,10>
    (tests/test_ekf.cpp:93-103) and test_predict_linear<9> (:148-153).  Bit-identical to the oracle: Euler and
    runge_kutta4 predict, update, fused predict + update, shared and per-item Q / R / dt."""
    rng = np.random.default_rng(dof * 1000 + ny * 10 + B)
    P = _flat(_spd(rng, B, dof)); A = _flat(rng.uniform(-1, 1, (B, dof, dof)))
    Q = _flat(0.1 * np.tile(np.eye(dof), (B, 1, 1)) + 0.01 * rng.uniform(-1, 1, (B, dof, dof)))
    dt = rng.uniform(0.01, 0.05, B)
    H = _flat(rng.uniform(-1, 1, (B, ny, dof))); R = _flat(0.1 * np.tile(np.eye(ny), (B, 1, 1)) + 0.01 * _spd(rng, B, ny))
    r = rng.uniform(-1, 1, (B, ny))
    P1, _, _ = sfb.ekf_step_batch_host(P, dof, A=A, Q=Q, dt=dt)
    ref1 = oracle.ekf_predict_batch(A, Q, dt, P)
    assert np.array_equal(P1, ref1)
    P2, d2, i2 = sfb.ekf_step_batch_host(P1, dof, H=H, R=R, r=r)
    ref2, dref, iref = oracle.ekf_update_batch(H, R, r, ref1, dof)
    assert np.array_equal(i2, iref) and np.array_equal(P2, ref2) and np.array_equal(d2, dref)
    Qs, Rs = Q[0].copy(), R[0].copy()
    P3, d3, i3 = sfb.ekf_step_batch_host(P, dof, A=A, Q=Qs, dt=0.025, H=H, R=Rs, r=r)
    refp = oracle.ekf_predict_batch(A, Qs, 0.025, P)
    ref3, dref3, _ = oracle.ekf_update_batch(H, Rs, r, refp, dof)
    assert np.array_equal(P3, ref3) and np.array_equal(d3, dref3) and (i3 == 0).all()
    got = sfb.ekf_predict_batch_host(P, dof, A, Q, dt, stepper="rk4")
    assert np.array_equal(got, oracle.ekf_predict_batch(A, Q, dt, P, stepper="rk4"))


def test_update_linear_identities_reference_sizes(sfb):
    """tests/test_ekf.cpp:93-103: test_update_linear<3,3>, <10,3>, <3,10> -- the textbook linear Kalman update
    through the device path, tolerance 1e-6 as in the reference."""
    rng = np.random.default_rng(5)
    for Nx, Ny in ((3, 3), (10, 3), (3, 10)):
        B = 10
        P = np.stack([np.diag(rng.uniform(-1, 1, Nx) + 1.1) for _ in range(B)])
        H = rng.uniform(-1, 1, (B, Ny, Nx)); R = np.stack([np.diag(rng.uniform(-1, 1, Ny) + 1.1) for _ in range(B)])
        r = rng.uniform(-1, 1, (B, Ny))
        Pn, delta, info = sfb.ekf_step_batch_host(_flat(P), Nx, H=_flat(H), R=_flat(R), r=r)
        S = H @ P @ H.transpose(0, 2, 1) + R
        K = P @ H.transpose(0, 2, 1) @ np.linalg.inv(S)
        assert np.allclose(delta, np.einsum("bij,bj->bi", K, r), rtol=1e-6, atol=1e-10)
        assert np.allclose(Pn.reshape(B, Nx, Nx).transpose(0, 2, 1), (np.eye(Nx)[None] - K @ H) @ P, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("dof", [3, 6, 9])
def test_rk4_with_stage_linearisations(sfb, oracle, dof):
    """Dynamics that depend on t explicitly: the reference's cov_ode re-linearises at every runge_kutta4 stage time
    (ekf.hpp:84-89), so the step takes A(t), A(t + dt/2), A(t + dt).  Bit-identical to the oracle; equal to the
    single-A step when the three coincide; and for dP/dt = A(t) P + P A(t)' with A(t) = a(t) A0 (commuting) the
    step reproduces expm(s A0) P expm(s A0)', s = int a, to RK4 accuracy -- which the frozen-A step does not."""
    import scipy.linalg as sl
    rng = np.random.default_rng(40 + dof)
    B = 70
    P = _flat(_spd(rng, B, dof)); Q = np.zeros(dof * dof)
    A0 = rng.uniform(-1, 1, (B, dof, dof))
    t0, h = 0.3, 0.05
    a = lambda t: 1.0 + 4.0 * t
    A, Am, Ae = _flat(a(t0) * A0), _flat(a(t0 + h / 2) * A0), _flat(a(t0 + h) * A0)
    got = sfb.ekf_predict_batch_host(P, dof, A, Q, h, stepper="rk4", A_mid=Am, A_end=Ae)
    ref = oracle.ekf_predict_batch(A, Q, h, P, stepper="rk4", A_mid=Am, A_end=Ae)
    assert np.array_equal(got, ref)
    same = sfb.ekf_predict_batch_host(P, dof, A, Q, h, stepper="rk4", A_mid=A, A_end=A)
    assert np.array_equal(same, sfb.ekf_predict_batch_host(P, dof, A, Q, h, stepper="rk4"))
    s_int = h * (1.0 + 4.0 * (t0 + h / 2))
    frozen = sfb.ekf_predict_batch_host(P, dof, A, Q, h, stepper="rk4")
    Pm = P.reshape(B, dof, dof).transpose(0, 2, 1)
    exact = np.stack([sl.expm(s_int * A0[b]) @ Pm[b] @ sl.expm(s_int * A0[b]).T for b in range(B)])
    err_tv = np.abs(got.reshape(B, dof, dof).transpose(0, 2, 1) - exact).max()
    err_fr = np.abs(frozen.reshape(B, dof, dof).transpose(0, 2, 1) - exact).max()
    assert err_tv < 2e-3 and err_fr > 20 * err_tv, (err_tv, err_fr)


def test_full_size_properties(sfb, oracle):
    """BASELINE configs[4]: 1 048 576 filters, Dof 6 / Ny 3, fused predict+update.  Properties:
    (i) a permuted batch gives permuted results (lane/position independence), (ii) P stays
    symmetric positive definite and the update never increases trace(P) relative to the predicted
    covariance, (iii) a 4096-item slice is bit-identical to the oracle."""
    import torch
    rng = np.random.default_rng(0)
    B, n, m = 1 << 20, 6, 3
    P = _flat(_spd(rng, B, n)); A = _flat(rng.uniform(-1, 1, (B, n, n)))
    H = _flat(rng.uniform(-1, 1, (B, m, n))); r = rng.uniform(-1, 1, (B, m))
    Q, R = (0.1 * np.eye(n)).flatten(), (0.1 * np.eye(m)).flatten()
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    def run(idx):
        dP, dA, dH, dr = T(P[idx]), T(A[idx]), T(H[idx]), T(r[idx])
        dQ, dR, ddt = T(Q), T(R), T(np.array([0.025]))
        dd = torch.empty((len(idx), n), dtype=torch.float64, device=dev)
        sfb.ekf_predict_update_batch_device(len(idx), n, m, dA.data_ptr(), dQ.data_ptr(), 1, ddt.data_ptr(), 1,
                                            dH.data_ptr(), dR.data_ptr(), 1, dr.data_ptr(), dP.data_ptr(), dd.data_ptr(),
                                            stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return dP.cpu().numpy(), dd.cpu().numpy()
    ident = np.arange(B)
    P1, d1 = run(ident)
    perm = rng.permutation(B)
    P2, d2 = run(perm)
    assert np.array_equal(P2, P1[perm]) and np.array_equal(d2, d1[perm])
    Pm = P1.reshape(B, n, n)
    assert np.array_equal(Pm, Pm.transpose(0, 2, 1))
    sub = Pm[::257]
    assert np.linalg.eigvalsh(sub).min() > 0
    Ppred = oracle.ekf_predict_batch(A[:4096], Q, 0.025, P[:4096])
    ref, dref, _ = oracle.ekf_update_batch(H[:4096], R, r[:4096], Ppred, n)
    assert np.array_equal(P1[:4096], ref) and np.array_equal(d1[:4096], dref)
    tr_pred = Ppred.reshape(-1, n, n).trace(axis1=1, axis2=2)
    assert (ref.reshape(-1, n, n).trace(axis1=1, axis2=2) <= tr_pred + 1e-12).all()


def test_cpp_front_mirrors_reference_ekf_checks(sfb):
    """include/smooth_feedback_amd/ekf.hpp (EKF<G>::predict/update over the C-ABI): PredictTimeCut
    (tests/test_ekf.cpp:155-180), UpdateLinear (:50-103) and an SE2 predict/update/predict run."""
    import ctypes as C
    from examples import models_lib as M
    err = np.zeros(3)
    assert M.lib().sfbx_test_ekf(err.ctypes.data_as(C.c_void_p)) == 0
    assert err[0] < 1e-12 and err[1] < 1e-6 and err[2] < 1e-12, err


@pytest.mark.parametrize("dof", [2, 3, 4, 6, 9])
@pytest.mark.parametrize("B", [1, 65, 3000])
def test_rk4_predict_matches_oracle(sfb, oracle, dof, B):
    """sfb_ekf_predict_stepper_batch(SFB_EKF_RK4): same bits as the oracle's runge_kutta4 step, per-item and
    shared Q / dt, non-symmetric P and Q included; the Euler selector is the plain predict."""
    rng = np.random.default_rng(900 + dof * 10 + B)
    P = _flat(_spd(rng, B, dof) + 0.05 * rng.uniform(-1, 1, (B, dof, dof)))
    A = _flat(rng.uniform(-1, 1, (B, dof, dof)))
    Q = _flat(0.1 * np.tile(np.eye(dof), (B, 1, 1)) + 0.02 * rng.uniform(-1, 1, (B, dof, dof)))
    dt = rng.uniform(0.005, 0.1, B)
    got = sfb.ekf_predict_batch_host(P, dof, A, Q, dt, stepper="rk4")
    ref = oracle.ekf_predict_batch(A, Q, dt, P, stepper="rk4")
    assert np.array_equal(got, ref)
    got = sfb.ekf_predict_batch_host(P, dof, A, Q[0].copy(), 0.025, stepper="rk4")
    ref = oracle.ekf_predict_batch(A, Q[0].copy(), 0.025, P, stepper="rk4")
    assert np.array_equal(got, ref)
    got = sfb.ekf_predict_batch_host(P, dof, A, Q, dt, stepper="euler")
    assert np.array_equal(got, oracle.ekf_predict_batch(A, Q, dt, P))


def test_cpp_front_predict_linear_with_rk4(sfb):
    """tests/test_ekf.cpp:104-153 through EKF<R^Nx, RK4> (include/smooth_feedback_amd/ekf.hpp): 700 substeps of
    1e-3, estimate and covariance against expm.  The reference asserts 1e-3 relative."""
    import ctypes as C
    import scipy.linalg as sl
    from examples import models_lib as M
    rng = np.random.default_rng(77)
    A3, A6 = rng.uniform(-1, 1, (3, 3)), rng.uniform(-1, 1, (6, 6))
    F3, F6 = sl.expm(0.7 * A3), sl.expm(0.7 * A6)
    err = np.zeros(2)
    cm = lambda a: np.ascontiguousarray(a.flatten("F"))
    bufs = [cm(A3), cm(F3), cm(A6), cm(F6)]
    rc = M.lib().sfbx_test_ekf_predict_linear(*[b.ctypes.data_as(C.c_void_p) for b in bufs], err.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert err[0] < 1e-6 and err[1] < 1e-6, err
    A9 = rng.uniform(-1, 1, (9, 9))       # test_predict_linear<9> (:152): beyond the register-resident kernels
    bufs = [cm(A9), cm(sl.expm(0.7 * A9))]
    rc = M.lib().sfbx_test_ekf_predict_linear9(*[b.ctypes.data_as(C.c_void_p) for b in bufs], err.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert err[0] < 1e-6 and err[1] < 1e-6, err


@pytest.mark.parametrize("dof,ny", [(6, 3), (3, 2), (9, 3), (3, 10)])
def test_non_finite_inputs_follow_the_reference_semantics(sfb, oracle, dof, ny):
    """NaN / inf / singular innovation covariances: the reference's EKF does not validate anything -- the values go
    through Eigen's LDLT (a NaN on the diagonal stays the pivot, an invalid pivot over a non-zero column is
    info() != Success, a zero pivot divides to 0) and the products.  Per-lane and generic kernels both give the
    oracle's P, delta and info, bit for bit (NaN payloads aside); the other filters of the batch are unaffected."""
    B = 16
    rng = np.random.default_rng(dof * 31 + ny)
    P = _flat(_spd(rng, B, dof)); A = _flat(rng.uniform(-1, 1, (B, dof, dof)))
    Q = _flat(0.1 * np.tile(np.eye(dof), (B, 1, 1))); dt = np.full(B, 0.02)
    H = _flat(rng.uniform(-1, 1, (B, ny, dof))); R = _flat(0.1 * np.tile(np.eye(ny), (B, 1, 1)) + 0.01 * _spd(rng, B, ny))
    r = rng.uniform(-1, 1, (B, ny))
    clean = (P.copy(), A.copy(), Q.copy(), H.copy(), R.copy(), r.copy())
    P[1, 0] = np.nan              # NaN on the covariance diagonal
    A[2, 1] = np.nan
    H[3, :] = 0.0; R[3, :] = 0.0  # S = 0: the all-zero branch of the factorisation
    R[4, 0] = np.nan              # NaN at the first pivot of S
    R[5, -1] = np.nan             # NaN at the last diagonal entry of S
    r[6, 0] = np.inf
    H[7, 0] = np.inf
    R[8, :] = 0.0; H[8, :] = 0.0; H[8, 0] = 1e-200   # S tiny: |d| <= DBL_MIN divides to 0
    P1, d1, i1 = sfb.ekf_step_batch_host(P, dof, A=A, Q=Q, dt=dt, H=H, R=R, r=r)
    refp = oracle.ekf_predict_batch(A, Q, dt, P)
    ref, dref, iref = oracle.ekf_update_batch(H, R, r, refp, dof)
    assert np.array_equal(i1, iref), (i1, iref)
    assert np.array_equal(P1, ref, equal_nan=True) and np.array_equal(d1, dref, equal_nan=True)
    Pc, dc, ic = sfb.ekf_step_batch_host(clean[0], dof, A=clean[1], Q=clean[2], dt=dt, H=clean[3], R=clean[4], r=clean[5])
    for b in (0, 9, 15):
        assert np.array_equal(Pc[b], P1[b]) and np.array_equal(dc[b], d1[b])
