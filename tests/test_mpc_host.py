"""Host side of the MPC path (C++ front in include/smooth_feedback_amd/, exercised through the example
harness): Lie-group identities, LGR mesh properties (reference tests/test_collocation_mesh.cpp), problem
sizes, and the transcription values against an independent numpy restatement of
ocp_to_qp_update_dyn/cr/ce (ocp_to_qp.hpp:198-373) for the SE2xR3 vehicle.  CPU only."""
import ctypes as C

import numpy as np

from examples import models_lib as M


def test_lie_group_identities():
    assert M.lib().sfbx_lie_selftest() < 1e-6


def test_mesh_constants_and_properties():
    nodes, w, D = M.mesh(1, 4)
    # LGR K=4 constants, mapped from [0,1] to [-1,1]
    assert np.allclose(2 * nodes - 1, [-1, -0.575318923521694, 0.181066271118531, 0.822824080974592, 1], atol=1e-14)
    assert np.allclose(2 * w, [0.125, 0.657688639960120, 0.776386937686344, 0.440924422353536, 0], atol=1e-14)
    assert np.allclose(D[0], [-4.25, -0.911068799217, 0.217395730606, -0.086939176287], atol=1e-11)
    assert np.allclose(D[4], [-2, 0.982438978216, -1.254112540974, 3.577796011738], atol=1e-11)
    for n_iv, K in ((1, 4), (3, 4), (13, 4), (2, 5)):
        nodes, w, D = M.mesh(n_iv, K)
        assert abs(w.sum() - 1.0) < 1e-14                      # weights sum to one
        assert nodes[0] == 0.0 and nodes[-1] == 1.0 and np.all(np.diff(nodes) > 0)
        # D differentiates polynomials of degree <= K exactly on the reference interval [-1, 1]
        tau = np.concatenate([2 * M.mesh(1, K)[0] - 1])
        for deg in range(K + 1):
            p, dp = tau ** deg, deg * tau ** max(deg - 1, 0) * (deg > 0)
            assert np.allclose(p @ D, dp[:K], atol=1e-11)
    n1, _, _ = M.mesh(2, 4)
    assert np.allclose(n1[4:8] - n1[0:4], 0.5)                 # interval nodes shift by the interval length


def test_problem_sizes_match_the_baseline_config():
    d = M.mpc_dims(12, 50)   # BASELINE configs[2]: nx=12, nu=2, K=50 -> 13 intervals x 4 nodes
    assert (d["N"], d["n"], d["m"]) == (52, 740, 740)
    d6 = M.mpc_dims(6, 50)
    assert (d6["N"], d6["n"], d6["m"]) == (52, 422, 422)
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(12, 50)
    assert d["nnzA"] == 624 * 18 + 104 * 14 + 12 * 12          # dyn rows Nx+Ki+Nu, cr rows Nx+Nu, ce rows Nx
    assert np.all(np.diff(Ap) > 0) and Aj.max() < d["n"]
    for r in range(d["m"]):                                     # strictly ascending columns per row
        assert np.all(np.diff(Aj[Ap[r]:Ap[r + 1]]) > 0)
    # P: diagonal, w_i*tf*Q on x_i (+ 0.5*Qtf on x_0), w_i*tf*R on u_i, no entry for x_N
    assert d["nnzP"] == 12 * 52 + 2 * 52
    nodes, w, _ = M.mesh(13, 4)
    assert np.allclose(Pv[:12], w[0] * 5.0 + 0.5) and np.allclose(Pv[12:24], w[1] * 5.0)
    st = M.mpc_stage(12, 50)
    assert (st > 0).sum() == 12 * 14 + 12                       # separators + the rows pinning x_0 are held back
    sep = st[np.arange(14) * 4 * 12]                            # ... in nested-dissection order of the chain of
    assert sep.max() == 4 and (sep == 4).sum() == 1             # 14 separators: one root, levels 1..4
    assert all(sep[i] != sep[i + 1] for i in range(13))         # neighbours never share a level


# ---- independent numpy restatement of the transcription for the vehicle (variant 6) ----
def _se2_exp(a):
    th = a[2]
    A = np.sin(th) / th if abs(th) > 1e-5 else 1 - th * th / 6
    B = (1 - np.cos(th)) / th if abs(th) > 1e-5 else th / 2
    return np.array([[np.cos(th), -np.sin(th), A * a[0] - B * a[1]], [np.sin(th), np.cos(th), B * a[0] + A * a[1]],
                     [0, 0, 1]])


def _se2_log(T):
    th = np.arctan2(T[1, 0], T[0, 0])
    A = np.sin(th) / th if abs(th) > 1e-5 else 1 - th * th / 6
    B = (1 - np.cos(th)) / th if abs(th) > 1e-5 else th / 2
    V = np.array([[A, -B], [B, A]])
    v = np.linalg.solve(V, T[:2, 2])
    return np.array([v[0], v[1], th])


def _ad(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [0, 0, 0]])


def _dr_expinv(a):
    th = a[2]
    k = 1 / 12 if abs(th) < 1e-4 else 1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))
    A = _ad(a)
    return np.eye(3) + 0.5 * A + k * A @ A


def test_transcription_values_match_numpy_restatement():
    variant, K, tf = 6, 50, 5.0
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K, tf)
    B = 5
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=11, tf=tf)
    nodes, w, D = M.mesh(13, 4)
    Nx, Nu, N = 6, 2, 52
    vdes = np.array([1.0, 0.0, 0.4])
    g0 = np.array([[0, -1, 2.5], [1, 0, 0], [0, 0, 1.0]])     # SE2(SO2(pi/2), (2.5, 0))
    dfdx = np.zeros((6, 6)); dfdx[0, 3] = dfdx[1, 4] = dfdx[2, 5] = 1; dfdx[3, 3] = -0.2; dfdx[5, 5] = -0.4
    dfdu = np.zeros((6, 2)); dfdu[3, 0] = dfdu[5, 1] = 1
    import random
    for b in range(B):
        t = 0.025 * (b % 400)
        A = np.zeros((d["m"], d["n"])); lo = np.zeros(d["m"]); hi = np.zeros(d["m"])
        for s in range(13):
            Mn, alpha = 4 * s, 2.0 * 13
            for i in range(4):
                node = Mn + i
                f = np.array([1.0, 0.0, 0.4, -0.2 * 1.0, 0.0, -0.4 * 0.4])      # f(xdes, udes = 0)
                dxl = np.array([1.0, 0.0, 0.4, 0, 0, 0])
                adm = np.zeros((6, 6)); adm[:3, :3] = _ad((f + dxl)[:3])
                rows = slice(node * Nx, (node + 1) * Nx)
                A[rows, node * Nx:(node + 1) * Nx] += tf * dfdx - tf / 2 * adm
                A[rows, Nx * (N + 1) + node * Nu:Nx * (N + 1) + (node + 1) * Nu] += tf * dfdu
                for j in range(5):
                    A[rows, (Mn + j) * Nx:(Mn + j + 1) * Nx] -= alpha * D[j, i] * np.eye(Nx)
                lo[rows] = hi[rows] = -tf * (f - dxl)
        for node in range(N):
            rows = slice(Nx * N + node * 2, Nx * N + node * 2 + 2)
            A[rows, Nx * (N + 1) + node * Nu:Nx * (N + 1) + (node + 1) * Nu] = np.eye(2)
            lo[rows], hi[rows] = -0.5, 0.5
        # ce rows: e = xdes(t) (-) x_b; the harness draws xi_b from std::mt19937_64 -> recover e from l
        e = -l[b, Nx * N + 2 * N:]
        J = np.eye(6); J[:3, :3] = _dr_expinv(e[:3])
        A[Nx * N + 2 * N:, :Nx] = J
        lo[Nx * N + 2 * N:] = hi[Nx * N + 2 * N:] = -e
        Ad = np.zeros_like(A)
        for r in range(d["m"]):
            Ad[r, Aj[Ap[r]:Ap[r + 1]]] = Av[b, Ap[r]:Ap[r + 1]]
        assert np.allclose(Ad, A, atol=1e-12), np.abs(Ad - A).max()
        assert np.allclose(l[b], lo, atol=1e-12) and np.allclose(u[b], hi, atol=1e-12)
        assert np.abs(e).max() <= 0.5 + 0.3      # the perturbation really is U(-0.5, 0.5)^6 (log of a nearby element)


def test_exact_parabola_satisfies_the_transcription():
    """The reference's only numeric pin on ocp_to_qp (tests/test_ocp_to_qp.cpp:84-106): for the double integrator on
    Mesh<5,5> refined to two intervals, tf = 2, linearised around xl(t) = (0.05 t^2, 0.1 t), ul = 0.1, the exact
    trajectory x(t) = (3 - 0.3 t + 0.05 t^2, -0.3 + 0.1 t), u = 0.1 satisfies l <= A var <= u to 1e-8: LGR
    collocation differentiates a quadratic exactly.  Same rows through the C++ MPC front (error coordinates)."""
    import ctypes as C
    out = np.zeros(6)
    assert M.lib().sfbx_test_ocp_to_qp_parabola(out.ctypes.data_as(C.c_void_p)) == 0
    lo, hi, N, n, m, nivals = out
    assert (N, nivals) == (10, 2) and n == 2 * 11 + 10 and m == 2 * 10 + 10 + 2
    assert lo >= -1e-8 and hi >= -1e-8, (lo, hi)


def test_generic_ocp_to_qp_reproduces_the_reference_test():
    """tests/test_ocp_to_qp.cpp:41-107 (OcpToQp.Basic) as written there, through the generic front ocp_to_qp()
    (include/smooth_feedback_amd/ocp_to_qp.hpp, reached by the reference's include path <smooth/feedback/ocp_to_qp.hpp>):
    sizes of P, q, A, l, u agree (:81-86) and the exact trajectory, stacked as [x_0 .. x_N | u_0 .. u_{N-1}], satisfies
    l <= A var <= u to 1e-8 (:105-106).  Cost entries: theta = |xf|^2 + 2 q, g = u^2 give P(x_N, x_N) = 1/2 d2theta = 1,
    P(u_i, u_i) = dtheta/dq * w_i tf * d2g = 4 w_i tf and q(u_i) = dtheta/dq * w_i tf * dg/du = 0.4 w_i tf at ul = 0.1
    (ocp_to_qp.hpp:172-194); derivatives by finite differences here."""
    o = M.ocp_to_qp_basic()
    N = 10
    n, m = 2 * (N + 1) + N, 2 * N + N + 2
    assert tuple(o[:7]) == (n, m, n, m, m, n, m)
    assert o[7] >= -1e-8 and o[8] >= -1e-8, (o[7], o[8])
    assert abs(o[9] - 1.0) < 1e-4 and abs(o[10] - 4.0) < 1e-4 and abs(o[11] - 0.4) < 1e-6


def test_time_concept_relative_setters_weights_and_structure_refresh():
    """MPC API beyond operator() (reference mpc.hpp:520-603, time.hpp:25-89), through the C++ front:
    set_xdes_rel / set_udes_rel reproduce the absolute-time setters (the relative version differentiates x(t) by central
    differences: agreement to 1e-6); the controller on a std::chrono clock assembles the same QP as with double seconds;
    set_weights stores the weights without re-transcribing P (the reference's v1 behaviour) while the constructor does
    transcribe them; a new desired trajectory after the analysis is looked at lazily and never destroys a plan that a
    device-resident swarm has pinned."""
    import ctypes as C
    out = np.full(9, -1.0)
    assert M.lib().sfbx_test_mpc_time_and_setters(out.ctypes.data_as(C.c_void_p)) == 0
    assert 0 <= out[0] <= 1e-6, out[0]
    assert 0 <= out[1] <= 1e-6, out[1]
    assert list(out[2:]) == [1.0] * 7, out


# ---- a SECOND, independently written restatement at the headline model (variant 12: SE2 x R3 x SE2 x R3, nx = 12, nu = 2,
# K = 50 -> n = m = 740): everything from the formulas of SURVEY.md section 8-a11 -- the LGR differentiation matrix from the
# Legendre polynomials, the agents' states from a Python mt19937_64 (the harness draws xi_b ~ U(-0.5, 0.5)^12 from
# std::mt19937_64(seed + b) and sets x_b = xdes(t_b) (+) xi_b), the rows of ocp_to_qp_update_dyn / _cr / _ce
# (ocp_to_qp.hpp:198-373, mpc.hpp:288-301) -- without the product's lie.hpp / mesh.hpp.  It does not pin the reference
# (which holds no numeric test of the transcription); it halves the chance that the one restatement is wrong. ----
class _MT64:
    """std::mt19937_64 (Matsumoto-Nishimura 2004, the standard's parameters)."""
    NN, MM, MASK = 312, 156, (1 << 64) - 1

    def __init__(self, seed):
        self.mt = [seed & self.MASK]
        for i in range(1, self.NN):
            self.mt.append((6364136223846793005 * (self.mt[-1] ^ (self.mt[-1] >> 62)) + i) & self.MASK)
        self.i = self.NN

    def __call__(self):
        if self.i >= self.NN:
            mt, NN, MM = self.mt, self.NN, self.MM
            for k in range(NN):
                x = (mt[k] & 0xFFFFFFFF80000000) | (mt[(k + 1) % NN] & 0x7FFFFFFF)
                mt[k] = mt[(k + MM) % NN] ^ (x >> 1) ^ (0xB5026F5AA96619E9 if x & 1 else 0)
            self.i = 0
        x = self.mt[self.i]
        self.i += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x & self.MASK


def _uniform(rng, a, b):
    """std::uniform_real_distribution<double>(a, b) of libstdc++: generate_canonical with ONE 64-bit draw."""
    return float(rng()) / 18446744073709551616.0 * (b - a) + a


def _lgr_diffmat(K):
    """LGR collocation on [-1, 1]: the K roots of P_{K-1} + P_K (the first one is -1), the basis = those nodes and +1;
    D[j, i] = l_j'(tau_i).  Scaled to the unit interval by the caller."""
    from numpy.polynomial import legendre as L
    c = np.zeros(K + 1); c[K - 1] = 1; c[K] = 1
    tau = np.sort(L.legroots(c).real)
    pts = np.concatenate([tau, [1.0]])
    D = np.zeros((K + 1, K))
    for j in range(K + 1):
        others = np.delete(pts, j)
        lj = np.poly(others) / np.prod(pts[j] - others)        # Lagrange basis polynomial through the K + 1 points
        D[j] = np.polyval(np.polyder(lj), tau)
    return tau, D


def test_headline_model_transcription_matches_a_second_numpy_restatement():
    variant, K, tf, B, seed = 12, 50, 5.0, 64, 1234567
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K, tf)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=seed, tf=tf)
    Nx, Nu, N, niv, Kn = 12, 2, 52, 13, 4
    assert (d["n"], d["m"]) == (Nx * (N + 1) + Nu * N, Nx * N + Nu * N + Nx) == (740, 740)
    tau, Dref = _lgr_diffmat(Kn)                                # reference interval [-1, 1]
    alpha = 2.0 * niv                                           # d/dt on [0, 1] split into 13 intervals = (2 * 13) d/dtau
    # the model (examples/vehicle_model.h, from mpc_asif_vehicle.cpp:73-79): two SE2 x R3 vehicles driven by one input
    vd, wd = np.array([1.0, 0.0, 0.4]), np.array([0.8, 0.0, 0.3])
    f = np.concatenate([vd, [-0.2 * vd[0], 0.0, -0.4 * vd[2]], wd, [-0.3 * wd[0], 0.0, -0.5 * wd[2]]])   # f(xdes, udes = 0)
    dxl = np.concatenate([vd, np.zeros(3), wd, np.zeros(3)])                                            # d/dt xdes (body velocity)
    dfdx = np.zeros((12, 12)); dfdu = np.zeros((12, 2))
    dfdx[0, 3] = dfdx[1, 4] = dfdx[2, 5] = 1; dfdx[3, 3] = -0.2; dfdx[5, 5] = -0.4
    dfdx[6, 9] = dfdx[7, 10] = dfdx[8, 11] = 1; dfdx[9, 9] = -0.3; dfdx[11, 11] = -0.5
    dfdu[3, 0] = dfdu[5, 1] = dfdu[9, 0] = dfdu[11, 1] = 1
    adm = np.zeros((12, 12)); adm[0:3, 0:3] = _ad((f + dxl)[0:3]); adm[6:9, 6:9] = _ad((f + dxl)[6:9])   # ad of the bundle: per part, 0 on R3
    xcol = lambda i: slice(i * Nx, (i + 1) * Nx)
    ucol = lambda i: slice(Nx * (N + 1) + i * Nu, Nx * (N + 1) + (i + 1) * Nu)
    # rows that do not depend on the agent: dynamics defects (:240-275) and the input box (:300-330)
    A0 = np.zeros((d["m"], d["n"])); lo0 = np.zeros(d["m"]); hi0 = np.zeros(d["m"])
    for s in range(niv):
        for i in range(Kn):
            node = s * Kn + i
            rows = slice(node * Nx, (node + 1) * Nx)
            A0[rows, xcol(node)] += tf * dfdx - 0.5 * tf * adm                 # tf (df/dx - 1/2 ad(f + dxl))
            A0[rows, ucol(node)] += tf * dfdu
            for j in range(Kn + 1):
                A0[rows, xcol(s * Kn + j)] -= alpha * Dref[j, i] * np.eye(Nx)  # - sum_j D_ji x_j
            lo0[rows] = hi0[rows] = -tf * (f - dxl)
    for node in range(N):
        rows = slice(Nx * N + node * Nu, Nx * N + (node + 1) * Nu)
        A0[rows, ucol(node)] = np.eye(Nu)
        lo0[rows], hi0[rows] = -0.5, 0.5
    ce = slice(Nx * N + Nu * N, d["m"])
    worst = 0.0
    for b in range(B):
        rng = _MT64(seed + b)
        xi = np.array([_uniform(rng, -0.5, 0.5) for _ in range(12)])
        # x_b = xdes (+) xi  =>  e = xdes (-) x_b = log(exp(xi)^-1) = -xi, part by part (SE2: exact in exact arithmetic)
        e = -xi
        J = np.eye(12); J[0:3, 0:3] = _dr_expinv(e[0:3]); J[6:9, 6:9] = _dr_expinv(e[6:9])   # d/dx of the (-) : dr_expinv per SE2 part
        A = A0.copy(); lo = lo0.copy(); hi = hi0.copy()
        A[ce, xcol(0)] = J
        lo[ce] = hi[ce] = -e
        Ad = np.zeros_like(A)
        for r in range(d["m"]):
            Ad[r, Aj[Ap[r]:Ap[r + 1]]] = Av[b, Ap[r]:Ap[r + 1]]
        worst = max(worst, np.abs(Ad - A).max(), np.abs(l[b] - lo).max(), np.abs(u[b] - hi).max())
        assert np.abs(Ad - A).max() <= 1e-12, (b, np.abs(Ad - A).max())
        assert np.abs(l[b] - lo).max() <= 1e-12 and np.abs(u[b] - hi).max() <= 1e-12, b
    print("second restatement, 64 agents of the headline model: max |difference| over A, l, u =", worst)


def test_reference_shaped_mpc_front_host_side():
    """The host-only half of the reference-shaped `MPC<T, X, U, F, CR, Kmesh>` (examples/models.cpp::sfbx_test_mpc_front_host; the
    solves of tests/test_mpc.cpp run in tests/test_mpc_gpu.py::test_reference_shaped_mpc_caller_code): copies share the desired
    trajectories (mpc.hpp:407, 607-608) and own their QP and their (un-analysed) solver, Ncr is read off CR's result (mpc.hpp:383)."""
    out = np.full(7, np.nan)
    assert M.lib().sfbx_test_mpc_front_host(out.ctypes.data_as(C.c_void_p)) == 0
    assert out[0] > 1e-3 and out[1] == 0.0
    assert out[2] == 1.0 and out[3] == 1.0
    assert out[4] == 0.0
    assert out[5] == 232.0 and out[6] == 1.0
