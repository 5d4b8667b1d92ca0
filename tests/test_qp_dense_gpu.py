"""Parity of the HIP dense-QP kernel (through the C-ABI) with the CPU oracle.  Needs an MI355X.

Bar: status codes and iteration counts bit-exact; primal/dual within 1e-8 absolute of the oracle
(BASELINE.json north_star: 'matching reference primal/dual ... to 1e-8').  In practice the kernel
follows the oracle's operation order, so the tests also report whether results are bit-identical.
"""
import json
import os

import numpy as np
import pytest

from qp_cases import KNOWN_ANSWERS, as_batch, is_approx

pytestmark = pytest.mark.gpu
TOL = 1e-8
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_params(oracle, prm):
    return oracle.default_params(
        alpha=prm.alpha, rho=prm.rho, sigma=prm.sigma, scaling=int(prm.scaling), eps_abs=prm.eps_abs,
        eps_rel=prm.eps_rel, eps_primal_inf=prm.eps_primal_inf, eps_dual_inf=prm.eps_dual_inf,
        max_iter=-1 if prm.max_iter is None else prm.max_iter, stop_check_iter=prm.stop_check_iter,
        polish=int(prm.polish), polish_iter=prm.polish_iter, delta=prm.delta)


def _compare(r, ref, tol=TOL):
    assert np.array_equal(r.code, ref["code"]), np.nonzero(r.code != ref["code"])
    assert np.array_equal(r.iter, ref["iter"]), np.nonzero(r.iter != ref["iter"])
    fin = np.isfinite(ref["x"]).all(axis=1) & np.isfinite(ref["y"]).all(axis=1)
    scale = 1.0 + np.maximum(np.abs(ref["x"]).max(axis=1), np.abs(ref["y"]).max(axis=1))[fin]
    dx = np.abs(r.primal - ref["x"])[fin].max(axis=1) / scale
    dy = np.abs(r.dual - ref["y"])[fin].max(axis=1) / scale
    assert dx.max(initial=0) <= tol and dy.max(initial=0) <= tol, (dx.max(), dy.max())
    do = np.abs(r.objective - ref["obj"])[fin] / (1.0 + np.abs(ref["obj"])[fin])
    assert do.max(initial=0) <= 1e-7
    return bool(np.array_equal(r.primal, ref["x"]) and np.array_equal(r.dual, ref["y"]))


@pytest.mark.parametrize("name", sorted(KNOWN_ANSWERS))
def test_known_answers(sfb, oracle, name):
    """tests/test_qp.cpp:54-336 through the device path, incl. the hot start from own solution."""
    case = KNOWN_ANSWERS[name]
    P, q, A, l, u = as_batch(case)
    prm = sfb.QPSolverParams(max_iter=100000)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    code, primal, ptol, objv, otol = case[5:]
    assert int(r.code[0]) == code
    if primal is not None:
        assert is_approx(r.primal[0], primal, ptol)
    if objv is not None:
        assert abs(r.objective[0] - objv) <= otol
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm))
    _compare(r, ref)
    r2 = sfb.solve_qp_batch_host(P, q, A, l, u, prm, warm_x=r.primal, warm_y=r.dual)
    ref2 = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), warm_x=ref["x"],
                                       warm_y=ref["y"])
    assert int(r2.code[0]) == code
    _compare(r2, ref2)


def test_reference_style_api(sfb):
    """QuadraticProgram / solve_qp / QPSolver mirror (tests/test_qp.cpp BasicStatic, SolverAPI)."""
    pbm = sfb.QuadraticProgram(P=np.eye(2), q=np.array([-4, 0.25]), A=np.eye(2), l=np.array([-1., -1.]),
                               u=np.array([1., 1.]))
    sol = sfb.solve_qp(pbm, sfb.QPSolverParams(polish=True))
    assert sol.code == sfb.QPSolutionStatus.Optimal
    assert is_approx(sol.primal, [1, -0.25], 1e-4)
    assert abs(sol.objective - (0.5 - 4 - 1 / 32)) < 1e-4
    hs = sfb.solve_qp(pbm, sfb.QPSolverParams(), sol)
    assert hs.code == sfb.QPSolutionStatus.Optimal and is_approx(hs.primal, [1, -0.25], 1e-4)
    s1 = sfb.QPSolver(pbm)
    a = s1.solve(pbm)
    b = sfb.QPSolver(pbm).solve(pbm)
    assert np.array_equal(a.primal, b.primal) and s1.sol() is a


@pytest.mark.parametrize("tag", ["default", "bench"])
def test_golden_fixture(sfb, tag):
    """Committed golden vectors (tests/golden/make_golden.py; densities 0.05/0.3/1.0, n=10, m=20)."""
    g = np.load(os.path.join(GOLD, "qp_dense_random.npz"))
    kw = json.loads(str(g["params_json"]))[tag]
    prm = sfb.QPSolverParams(**{k: (bool(v) if k in ("polish", "scaling") else v) for k, v in kw.items()})
    r = sfb.solve_qp_batch_host(g["P"], g["q"], g["A"], g["l"], g["u"], prm)
    ref = {k: g["%s_%s" % (tag, k)] for k in ("code", "iter", "x", "y", "obj")}
    _compare(r, ref)


@pytest.mark.parametrize("density", [0.05, 0.3, 1.0])
@pytest.mark.parametrize("tag", ["default", "bench"])
def test_random_batches_n10_m20(sfb, oracle, density, tag):
    """BASELINE configs[1] shape (n=10, m=20) at oracle-sized batches."""
    B = 1024
    P, q, A, l, u = sfb.random_qp_batch(7, B, 20, 10, density)
    if tag == "bench":  # benchmarks/bench.cpp:148-153
        prm = sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=10000, scaling=False)
    else:
        prm = sfb.QPSolverParams(max_iter=20000)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
    bit = _compare(r, ref)
    print("density", density, tag, "bit-identical primal/dual:", bit, "codes", np.bincount(r.code, minlength=7))


@pytest.mark.parametrize("n,m", [(1, 1), (2, 3), (3, 13), (5, 11), (10, 6), (16, 16), (12, 28), (20, 28),
                                 (24, 40), (32, 32), (8, 56)])
def test_other_sizes(sfb, oracle, n, m):
    """every kernel specialisation (k<=16, <=32, <=48, <=64) incl. ragged sizes and k == 64"""
    B = 96
    P, q, A, l, u = sfb.random_qp_batch(11 + n, B, m, n, 0.7)
    # mix of constraint kinds: two-sided, equality, free rows (exercise all rho classes, qp_solver.hpp:366-373)
    rng = np.random.default_rng(n * 100 + m)
    l = np.where(rng.random((B, m)) < 0.4, u - rng.random((B, m)) * 2.0, l)
    eq = rng.random((B, m)) < 0.1
    l = np.where(eq, u, l)
    free = rng.random((B, m)) < 0.1
    l = np.where(free, -np.inf, l)
    u = np.where(free, np.inf, u)
    prm = sfb.QPSolverParams(max_iter=3000)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
    _compare(r, ref)


def test_no_polish_no_scaling_and_stop_check_variants(sfb, oracle):
    B = 256
    P, q, A, l, u = sfb.random_qp_batch(3, B, 20, 10, 1.0)
    for prm in (sfb.QPSolverParams(polish=False, max_iter=2000),
                sfb.QPSolverParams(scaling=False, max_iter=2000),
                sfb.QPSolverParams(stop_check_iter=1, max_iter=60),     # never checks: qp_solver.hpp:465
                sfb.QPSolverParams(stop_check_iter=7, max_iter=2000, alpha=1.0, rho=1.0),
                sfb.QPSolverParams(max_iter=0), sfb.QPSolverParams(max_iter=1), sfb.QPSolverParams(max_iter=2)):
        r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
        ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
        _compare(r, ref)


def test_warm_start_batch(sfb, oracle):
    B = 256
    P, q, A, l, u = sfb.random_qp_batch(9, B, 20, 10, 1.0)
    prm = sfb.QPSolverParams(max_iter=5000)
    op = _oracle_params(oracle, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=op, nthreads=8)
    wx = np.where(np.isfinite(ref["x"]), ref["x"], 0.0)
    wy = np.where(np.isfinite(ref["y"]), ref["y"], 0.0)
    q2 = q + 0.01
    r = sfb.solve_qp_batch_host(P, q2, A, l, u, prm, warm_x=wx, warm_y=wy)
    ref2 = oracle.qp_dense_solve_batch(P, q2, A, l, u, params=op, warm_x=wx, warm_y=wy, nthreads=8)
    _compare(r, ref2)


def test_empty_batch_and_device_pointer_api(sfb, oracle):
    import torch
    r = sfb.solve_qp_batch_host(np.zeros((0, 100)), np.zeros((0, 10)), np.zeros((0, 200)), np.zeros((0, 20)),
                                np.zeros((0, 20)))
    assert r.code.shape == (0,)
    B, m, n = 512, 20, 10
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(P=P, q=q, A=A, l=l, u=u).items()}
    x = torch.empty((B, n), dtype=torch.float64, device=dev)
    y = torch.empty((B, m), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    code = torch.empty(B, dtype=torch.int32, device=dev)
    prm = sfb.QPSolverParams(max_iter=4000)
    stream = torch.cuda.current_stream()
    sfb.solve_qp_batch_device(B, n, m, t["P"].data_ptr(), t["q"].data_ptr(), t["A"].data_ptr(), t["l"].data_ptr(),
                              t["u"].data_ptr(), x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(),
                              code.data_ptr(), prm, stream=stream.cuda_stream)
    stream.synchronize()
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
    assert np.array_equal(code.cpu().numpy(), ref["code"])
    assert np.array_equal(it.cpu().numpy().astype(np.uint32), ref["iter"])
    assert np.abs(x.cpu().numpy() - ref["x"]).max() <= TOL


def test_full_size_batch_properties(sfb):
    """BASELINE configs[1] at full size (65 536 QPs): size-independent properties instead of the
    oracle: (i) determinism / batch-position independence (a permuted batch gives permuted results),
    (ii) every Optimal solution satisfies the KKT residual bounds the stopping test promises."""
    B, m, n = 65536, 20, 10
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
    prm = sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=10000, scaling=False)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    perm = np.random.default_rng(0).permutation(B)
    r2 = sfb.solve_qp_batch_host(P[perm], q[perm], A[perm], l[perm], u[perm], prm)
    assert np.array_equal(r2.code, r.code[perm]) and np.array_equal(r2.iter, r.iter[perm])
    assert np.array_equal(r2.primal, r.primal[perm]) and np.array_equal(r2.dual, r.dual[perm])
    opt = r.code == 0
    assert opt.sum() > B // 4
    Pm = P.reshape(B, n, n).transpose(0, 2, 1)[opt]
    Am = A.reshape(B, n, m).transpose(0, 2, 1)[opt]
    x, y = r.primal[opt], r.dual[opt]
    Ax = np.einsum("bij,bj->bi", Am, x)
    assert (Ax - u[opt]).max() <= 1e-5                              # primal feasibility (l = -inf)
    res = np.einsum("bij,bj->bi", Pm, x) + q[opt] + np.einsum("bji,bj->bi", Am, y)
    assert np.abs(res).max() <= 1e-4                                # stationarity
    assert y.min() >= -1e-3                                         # dual sign (polished duals: approx.)
    assert np.abs(y * (Ax - u[opt])).max() <= 1e-4                  # complementarity


@pytest.fixture
def env_knob(knobs):
    """Debug knobs of libsfb.so for one test (conftest.knobs)."""
    return knobs.set


@pytest.mark.parametrize("waves", [1, 3])
@pytest.mark.parametrize("n,m", [(10, 20), (4, 9)])
def test_four_per_wave_slot_refill(sfb, oracle, env_knob, waves, n, m):
    """qp_dense4_kernel with a grid of 1 / 3 persistent waves: every slot is refilled many times from the
    device-side queue, at different iterations (the queue hands out problems of 2 ... max_iter iterations),
    for ragged batch sizes, aligned and unaligned max_iter and several stop_check_iter."""
    env_knob(SFB_QP4_MAX_WAVES=waves)
    for B, prm in ((1, sfb.QPSolverParams(max_iter=500)),
                   (2, sfb.QPSolverParams(max_iter=26)),
                   (3, sfb.QPSolverParams(max_iter=27)),
                   (5, sfb.QPSolverParams(max_iter=51, stop_check_iter=2)),
                   (61, sfb.QPSolverParams(max_iter=700)),
                   (97, sfb.QPSolverParams(max_iter=333, stop_check_iter=3, polish=False)),
                   (64, sfb.QPSolverParams(max_iter=40, stop_check_iter=1)),
                   (50, sfb.QPSolverParams(max_iter=300, stop_check_iter=1000)),
                   (131, sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, max_iter=1500, scaling=False))):
        P, q, A, l, u = sfb.random_qp_batch(100 + B, B, m, n, 0.8)
        l[::7] = -np.inf                       # one-sided rows
        l[3::11] = u[3::11]                    # equality rows
        if B > 10:
            l[5], u[5] = 1.0, -1.0             # u < l: PrimalInfeasible before the first iteration (:361-364)
        r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
        ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
        _compare(r, ref)
    # warm start through the refill path
    B = 77
    P, q, A, l, u = sfb.random_qp_batch(9, B, m, n, 1.0)
    prm = sfb.QPSolverParams(max_iter=2000)
    op = _oracle_params(oracle, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=op, nthreads=8)
    wx = np.where(np.isfinite(ref["x"]), ref["x"], 0.0)
    wy = np.where(np.isfinite(ref["y"]), ref["y"], 0.0)
    r = sfb.solve_qp_batch_host(P, q + 0.02, A, l, u, prm, warm_x=wx, warm_y=wy)
    ref2 = oracle.qp_dense_solve_batch(P, q + 0.02, A, l, u, params=op, warm_x=wx, warm_y=wy, nthreads=8)
    _compare(r, ref2)


@pytest.mark.parametrize("waves,slice_checks", [(1, 1), (5, 2), (16, 40)])
@pytest.mark.parametrize("n,m", [(16, 32), (20, 44), (40, 60), (64, 64), (100, 20)])
def test_mid_kernel_time_sliced_launch(sfb, oracle, env_knob, waves, slice_checks, n, m):
    """qp_dense_mid_kernel (32 < n+m <= 128) as a persistent grid of 1 / 5 / 16 waves with a slice of 1 / 2 / 40 check
    intervals: QPs are suspended (factor, permutation and iterate to the workspace) and resumed by other waves many
    times; results equal the oracle's bit for bit, as with one QP per workgroup -- cold and warm starts, infeasible
    items, max_iter on and off a check iteration, no checks at all."""
    env_knob(SFB_MID_GRID=waves, SFB_MID_SLICE=slice_checks)
    for B, prm in ((37, sfb.QPSolverParams(max_iter=700)),
                   (23, sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, max_iter=403, scaling=False)),
                   (19, sfb.QPSolverParams(max_iter=77, stop_check_iter=3, polish=False)),
                   (18, sfb.QPSolverParams(max_iter=130, stop_check_iter=0))):
        P, q, A, l, u = sfb.random_qp_batch(300 + B, B, m, n, 0.8)
        l[::7] = -np.inf
        l[3::11] = u[3::11]
        l[5], u[5] = 1.0, -1.0             # u < l: PrimalInfeasible before the first iteration (:361-364)
        r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
        ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
        _compare(r, ref)
    B = 29
    P, q, A, l, u = sfb.random_qp_batch(12, B, m, n, 1.0)
    prm = sfb.QPSolverParams(max_iter=900)
    op = _oracle_params(oracle, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=op, nthreads=8)
    wx = np.where(np.isfinite(ref["x"]), ref["x"], 0.0)
    wy = np.where(np.isfinite(ref["y"]), ref["y"], 0.0)
    r = sfb.solve_qp_batch_host(P, q + 0.02, A, l, u, prm, warm_x=wx, warm_y=wy)
    ref2 = oracle.qp_dense_solve_batch(P, q + 0.02, A, l, u, params=op, warm_x=wx, warm_y=wy, nthreads=8)
    _compare(r, ref2)


@pytest.mark.parametrize("n,m", [(40, 30), (64, 96), (3, 203), (4, 301), (33, 32), (120, 250)])
def test_larger_dense_problems_are_bit_identical_to_the_dense_oracle(sfb, oracle, n, m):
    """64 < n + m <= 1024 through the SAME dense entry point: the pivoted dense LDL' of the reference's dense branch
    (qp_solver.hpp:259,:428,:462) with the factor in the QP's HBM workspace (qp_dense_big.hip).  Codes, iteration
    counts, primal, dual and objective equal the dense oracle's bit for bit -- including the sizes of the
    reference's ASIF example (n = 3, m = 203) and test (n = 4, m = 301).  Cold and warm start."""
    B = 24 if n + m > 300 else 48
    P, q, A, l, u = sfb.random_qp_batch(200 + n, B, m, n, 0.6)
    rng = np.random.default_rng(n + m)
    l = np.where(rng.random((B, m)) < 0.5, u - 1.0 - rng.random((B, m)), l)
    if m > 100:   # wide bands around the generator's point v (u = A v + delta): feasible, few active rows -- like the
        u = u + 5.0   # barrier constraints of a safety filter
        l = np.where(np.isfinite(l), l - 5.0, l)
    prm = sfb.QPSolverParams(max_iter=4000)
    op = _oracle_params(oracle, prm)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=op, nthreads=8)
    for a, b in ((r.code, ref["code"]), (r.iter, ref["iter"]), (r.primal, ref["x"]), (r.dual, ref["y"]), (r.objective, ref["obj"])):
        assert np.array_equal(a, b, equal_nan=True)
    assert (r.code == 0).sum() >= B // 3
    r2 = sfb.solve_qp_batch_host(P, q + 0.01, A, l, u, prm, warm_x=r.primal, warm_y=r.dual)
    ref2 = oracle.qp_dense_solve_batch(P, q + 0.01, A, l, u, params=op, warm_x=ref["x"], warm_y=ref["y"], nthreads=8)
    for a, b in ((r2.code, ref2["code"]), (r2.iter, ref2["iter"]), (r2.primal, ref2["x"]), (r2.dual, ref2["y"])):
        assert np.array_equal(a, b, equal_nan=True)
    # no scaling, tight tolerances, no polish (the benchmark's and the ASIF example's parameter styles)
    prm3 = sfb.QPSolverParams(max_iter=2000, scaling=False, eps_abs=1e-6, eps_rel=1e-6, polish=False)
    r3 = sfb.solve_qp_batch_host(P[:8], q[:8], A[:8], l[:8], u[:8], prm3)
    ref3 = oracle.qp_dense_solve_batch(P[:8], q[:8], A[:8], l[:8], u[:8], params=_oracle_params(oracle, prm3), nthreads=8)
    for a, b in ((r3.code, ref3["code"]), (r3.iter, ref3["iter"]), (r3.primal, ref3["x"]), (r3.dual, ref3["y"])):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("n,m,B", [(20, 40, 4096), (40, 60, 4096), (64, 64, 4096), (3, 203, 4096)])
def test_bench_batch_parity_of_the_mid_and_big_routes(sfb, oracle, n, m, B):
    """The dense routes beyond the four-per-wave kernel at a batch larger than the chip, on the benchmark's generator
    (benchmarks/bench_types.hpp:19-41) under the reference benchmark's parameters (bench.cpp:148-153) and under the
    library defaults: the split, time-sliced launch of qp_dense_mid (k = 60: one block per wave and two waves per SIMD;
    k = 100, 128: two rows per lane) and the pivoted big kernel ((3, 203)) against the oracle on EVERY QP of the batch --
    codes, iteration counts, primal, dual, objective bit for bit.  The suite's other tests use 24-96 QPs per route."""
    import os
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
    if m > 128:   # a safety filter's shape: bands around the generator's point, few active rows
        u = u + 5.0
    for prm in (sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=10000 if n + m <= 128 else 1500, scaling=False),
                sfb.QPSolverParams(max_iter=10000 if n + m <= 128 else 1500)):
        r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
        ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=min(64, os.cpu_count() or 8))
        assert int((r.code != ref["code"]).sum()) == 0 and int((r.iter != ref["iter"]).sum()) == 0
        for a, b in ((r.primal, ref["x"]), (r.dual, ref["y"]), (r.objective, ref["obj"])):
            assert np.array_equal(a, b, equal_nan=True)


def test_known_answers_padded_beyond_64(sfb, oracle):
    """The reference's known answers (tests/test_qp.cpp) embedded in larger problems (extra free rows and
    decoupled variables push n + m past 64): the big dense kernel reproduces codes and solutions."""
    for name, case in sorted(KNOWN_ANSWERS.items()):
        P0, q0, A0, l0, u0 = (np.asarray(t, dtype=np.float64) for t in case[:5])
        n0, m0 = len(q0), len(l0)
        n, m = n0 + 30, m0 + 40
        P = np.eye(n); P[:n0, :n0] = P0
        q = np.zeros(n); q[:n0] = q0
        A = np.zeros((m, n)); A[:m0, :n0] = A0
        for e in range(30):
            A[m0 + e, n0 + e] = 1.0            # box rows on the extra variables
        l = np.full(m, -np.inf); u = np.full(m, np.inf)
        l[:m0], u[:m0] = l0, u0
        l[m0:m0 + 30], u[m0:m0 + 30] = -1.0, 1.0
        prm = sfb.QPSolverParams(max_iter=100000)
        Pb, Ab = np.ascontiguousarray(P.T.reshape(1, -1)), np.ascontiguousarray(A.T.reshape(1, -1))
        r = sfb.solve_qp_batch_host(Pb, q[None], Ab, l[None], u[None], prm)
        ref = oracle.qp_dense_solve_batch(Pb, q[None], Ab, l[None], u[None], params=_oracle_params(oracle, prm))
        code, primal, ptol = case[5], case[6], case[7]
        assert int(r.code[0]) == code == int(ref["code"][0]), name
        assert np.array_equal(r.iter, ref["iter"]) and np.array_equal(r.primal, ref["x"], equal_nan=True)
        if primal is not None:
            assert is_approx(r.primal[0, :n0], primal, ptol), name


def test_sizes_beyond_the_big_dense_kernel_use_the_sparse_kernel(sfb, oracle, env_knob):
    """n + m > 1024 (and, with SFB_QP_DENSE_BIG=0, everything above 64) runs on the shared-pattern sparse kernel with
    a full pattern: a fill-reducing order without pivoting -- same codes, iteration counts within one check
    interval, solutions to tolerance."""
    env_knob(SFB_QP_DENSE_BIG=0)
    n, m, B = 40, 30, 16
    P, q, A, l, u = sfb.random_qp_batch(240, B, m, n, 0.6)
    prm = sfb.QPSolverParams(max_iter=4000)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8)
    assert np.array_equal(r.code, ref["code"])
    assert np.abs(r.iter.astype(np.int64) - ref["iter"].astype(np.int64)).max() <= 25
    same = (r.iter == ref["iter"]) & (ref["code"] == 0)
    assert same.sum() >= B // 2
    scale = 1.0 + np.abs(ref["x"][same]).max(axis=1)
    assert (np.abs(r.primal[same] - ref["x"][same]).max(axis=1) / scale).max() <= 1e-6


def test_cpp_front_solver_api_like_the_reference(sfb):
    """tests/test_qp.cpp StaticProperties, SolverAPI, SparseSolverAPI, PartialDynamic through the C++ front, spelled
    with the reference's include path and namespace (<smooth/feedback/qp_solver.hpp>, smooth::feedback::QPSolver<Pbm>):
    copies and moved-to solvers give the same primal; dense == sparse (TwoDimensional, :314-336)."""
    import ctypes as C
    from examples import models_lib as M
    pd, ps, pp = np.zeros((5, 2)), np.zeros((5, 2)), np.zeros(5)
    rc = M.lib().sfbx_test_qp_solver_api(*[a.ctypes.data_as(C.c_void_p) for a in (pd, ps, pp)])
    assert rc == 0, rc
    for arr in (pd, ps):
        assert all(np.array_equal(arr[0], arr[i]) for i in range(1, 5))
    assert is_approx(pd[0], [46.6338, -17.5351], 1e-4)         # tests/test_qp.cpp:334-335
    assert is_approx(pd[0], ps[0], 1e-10)                      # :332-333 isApprox(dense, sparse)
    assert is_approx(pp[:2], [1, -0.25], 1e-4) and abs(pp[2] - (0.5 - 4 - 1 / 32)) < 1e-4 and is_approx(pp[3:], [1, -0.25], 1e-4)


@pytest.mark.parametrize("n,m", [(10, 20), (30, 30), (40, 60)])
def test_max_time_like_the_reference(sfb, oracle, n, m):
    """QPSolverParams::max_time (qp_solver.hpp:504-507): tested at stopping checks that leave the status open, on
    the device clock.  A limit of 1 ns makes it deterministic: every QP ends at its first check (iteration 2) with
    the status that check gives, else MaxTime -- same as the oracle with the same limit.  A generous limit changes
    nothing.  (k <= 32: the one-QP-per-wave kernels take over from the four-per-wave one when a limit is set.)"""
    B = 64
    P, q, A, l, u = sfb.random_qp_batch(77 + n, B, m, n, 0.8)
    prm = sfb.QPSolverParams(max_iter=4000, max_time=1e-9)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    op = _oracle_params(oracle, prm)
    op.max_time_ns = 1
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=op, nthreads=8)
    assert np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"])
    assert (r.code == sfb.QPSolutionStatus.MaxTime).sum() > B // 2 and set(np.unique(r.iter)) <= {0, 2}
    assert np.array_equal(r.primal, ref["x"], equal_nan=True)
    base = sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=4000))
    slow = sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=4000, max_time=30.0))
    assert np.array_equal(base.code, slow.code) and np.array_equal(base.iter, slow.iter) and np.array_equal(base.primal, slow.primal, equal_nan=True)


def test_verbose_prints_a_summary_of_the_call(sfb, capfd):
    """QPSolverParams::verbose (qp_solver.hpp:409-420, :550-565 print phase times and the outcome): the host-pointer
    entry points print one summary per call."""
    P, q, A, l, u = sfb.random_qp_batch(5, 32, 20, 10, 1.0)
    sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=500, verbose=True))
    out = capfd.readouterr().out
    assert "[sfb] dense QP batch: 32 problem(s), n = 10, m = 20" in out and "status:" in out and "iterations: min" in out
    sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=500))
    assert "[sfb]" not in capfd.readouterr().out


@pytest.mark.parametrize("n,m", [(3, 5), (10, 20), (20, 30), (40, 60), (3, 203)])
def test_explicit_workspace_gives_the_same_results(sfb, n, m):
    """sfb_workspace_create / sfb_qp_dense_solve_batch_ws: the caller's device workspace instead of stream-ordered
    allocations inside the call -- every dense path (four-per-wave, one-per-wave, big dense), same results bit for
    bit, twice on the same workspace; a workspace that is too small is refused."""
    import torch
    B = 96
    P, q, A, l, u = sfb.random_qp_batch(11, B, m, n, 0.6)
    prm = sfb.QPSolverParams(max_iter=600)
    ref = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    need = sfb.Workspace.dense_bytes(B, n, m, prm)
    assert (need == 0) == (32 < n + m <= 128)  # the on-chip kernels (qp_dense.hip, qp_dense_mid.hip) need none
    ws = sfb.Workspace(need)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (P, q, A, l, u)]
    for _ in range(2):
        x = torch.full((B, n), np.nan, dtype=torch.float64, device=dev); y = torch.full((B, m), np.nan, dtype=torch.float64, device=dev)
        obj = torch.full((B,), np.nan, dtype=torch.float64, device=dev)
        it = torch.zeros(B, dtype=torch.int32, device=dev); code = torch.full((B,), -1, dtype=torch.int32, device=dev)
        sfb.solve_qp_batch_device_ws(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(),
                                     it.data_ptr(), code.data_ptr(), ws, prm, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(code.cpu().numpy(), ref.code) and np.array_equal(it.cpu().numpy().astype(np.uint32), ref.iter)
        assert np.array_equal(x.cpu().numpy(), ref.primal, equal_nan=True) and np.array_equal(y.cpu().numpy(), ref.dual, equal_nan=True)
        assert np.array_equal(obj.cpu().numpy(), ref.objective, equal_nan=True)
    if need:
        small = sfb.Workspace(need - 8)
        with pytest.raises(sfb._capi.SfbError) as ei:
            sfb.solve_qp_batch_device_ws(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(),
                                         it.data_ptr(), code.data_ptr(), small, prm)
        assert ei.value.status == sfb._capi.SFB_ERR_INVALID_ARG


@pytest.mark.parametrize("n,m", [(10, 20), (20, 30), (40, 60)])
def test_non_finite_inputs_follow_the_reference_semantics(sfb, oracle, n, m):
    """NaN / inf in the problem data: the reference has no input validation -- a NaN in q, P or A simply propagates
    through the IEEE arithmetic (std::max-style norms ignore it, comparisons with it are false), l = +inf or u = -inf
    is the pre-check's PrimalInfeasible (qp_solver.hpp:361-374).  Every dense kernel (four-per-wave, one-per-wave, big)
    must give what the oracle gives: same codes and iteration counts, same primal / dual bit patterns up to NaN
    payloads; the clean items of the batch are unaffected."""
    B = 12
    P, q, A, l, u = sfb.random_qp_batch(3, B, m, n, 0.8)
    q[1, 0] = np.nan
    P[2, 0] = np.nan
    A[3, 1] = np.inf
    l[4, 2] = np.inf
    u[5, 3] = -np.inf
    l[6, 0] = np.nan
    q[7, :] = np.inf
    A[8, :] = 0.0
    prm = sfb.QPSolverParams(max_iter=300)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm))
    assert np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"]), (r.code, ref["code"], r.iter, ref["iter"])
    assert np.array_equal(r.primal, ref["x"], equal_nan=True) and np.array_equal(r.dual, ref["y"], equal_nan=True)
    assert r.code[4] == 2 and r.code[5] == 2 and r.iter[4] == 0
    clean = sfb.random_qp_batch(3, B, m, n, 0.8)
    rc = sfb.solve_qp_batch_host(*clean, prm)
    for b in (0, 9, 10, 11):
        assert np.array_equal(rc.primal[b], r.primal[b]) and rc.iter[b] == r.iter[b]


def test_verbose_single_dense_problem_prints_the_reference_table(sfb, capfd):
    """QPSolverParams::verbose on ONE dense problem (the reference's use: solve_qp / one QPSolver, qp_solver.hpp:409-420,
    :490-501): header, one line per stopping check with the residuals falling to the tolerances, then the summary; the
    results are those of the plain call."""
    P, q, A, l, u = sfb.random_qp_batch(3, 1, 20, 10, 1.0)
    plain = sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=4000))
    capfd.readouterr()
    r = sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=4000, verbose=True))
    out = capfd.readouterr().out
    assert np.array_equal(r.primal, plain.primal) and np.array_equal(r.iter, plain.iter) and np.array_equal(r.code, plain.code)
    assert "========================= QP Solver" in out and "Solving dense QP with n=10, m=20" in out
    rows = [ln.split(":")[1].split() for ln in out.splitlines() if ":" in ln and ln.split(":")[0].strip().isdigit()]
    its = [int(ln.split(":")[0]) for ln in out.splitlines() if ":" in ln and ln.split(":")[0].strip().isdigit()]
    assert its[0] == 1 and all(b - a == 25 for a, b in zip(its, its[1:]))
    if int(r.code[0]) == 0:  # Optimal: as many checks as the solve's own count says, the last row within the tolerances' range
        assert len(its) == (int(r.iter[0]) - 2) // 25 + 1
        assert float(rows[-1][1]) < float(rows[0][1]) or float(rows[-1][2]) < float(rows[0][2])
    # batches keep the summary only
    sfb.solve_qp_batch_host(*sfb.random_qp_batch(3, 4, 20, 10, 1.0), sfb.QPSolverParams(max_iter=100, verbose=True))
    out = capfd.readouterr().out
    assert "QP Solver ====" not in out and "[sfb] dense QP batch: 4 problem(s)" in out


@pytest.mark.parametrize("n,m,B", [(3, 5, 40), (10, 20, 64), (16, 32, 48), (20, 40, 48), (32, 64, 24), (40, 60, 24), (64, 64, 12)])
def test_verbose_table_as_data_matches_the_dense_oracle_trace(sfb, oracle, n, m, B):
    """qp_solver.hpp:490-501 for dense problems, natively: ITER, OBJ, PRI_RES, DUA_RES per stopping check from the TRACE
    instance of the on-chip dense kernel (n + m <= 128, every block count incl. the sizes the four-per-wave kernel
    normally takes) against the dense oracle's trace bit for bit, and the results of the traced call against the plain
    one -- the table is made of the iterates of the very solve whose results are returned."""
    rows = 10
    P, q, A, l, u = sfb.random_qp_batch(23, B, m, n, 0.9)
    prm = sfb.QPSolverParams(max_iter=202)
    plain = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm, trace_rows=rows)
    assert np.array_equal(r.code, plain.code) and np.array_equal(r.iter, plain.iter)
    assert np.array_equal(r.primal, plain.primal, equal_nan=True) and np.array_equal(r.dual, plain.dual, equal_nan=True)
    assert np.array_equal(r.objective, plain.objective, equal_nan=True)
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, prm), nthreads=8, trace_rows=rows)
    assert np.array_equal(r.iter, ref["iter"]) and np.array_equal(r.code, ref["code"])
    tr, tref = r.trace, ref["trace"]
    assert np.array_equal(tr[:, :, 0], tref[:, :, 0]), "check iterations differ"
    used = tref[:, :, 0] >= 0
    assert used.sum() > B and (tr[:, :, 0][used] % 25 == 1).all()
    for col, name in ((1, "OBJ"), (2, "PRI_RES"), (3, "DUA_RES")):
        a, b = tr[:, :, col][used], tref[:, :, col][used]
        fin = np.isfinite(b)
        assert np.array_equal(a[fin], b[fin]), (name, np.abs(a[fin] - b[fin]).max())
    both = used[:, 1:] & used[:, :-1]
    assert (tr[:, :, 4][used] >= 0).all() and (tr[:, 1:, 4][both] >= tr[:, :-1, 4][both]).all()  # TIME grows
    # warm start and a table shorter than the number of checks
    w = sfb.solve_qp_batch_host(P, q, A, l, u, prm, warm_x=0.5 * plain.primal, warm_y=0.5 * plain.dual, trace_rows=2)
    wp = sfb.solve_qp_batch_host(P, q, A, l, u, prm, warm_x=0.5 * plain.primal, warm_y=0.5 * plain.dual)
    assert w.trace.shape == (B, 2, 5) and np.array_equal(w.iter, wp.iter) and np.array_equal(w.primal, wp.primal, equal_nan=True)


def test_dense_trace_is_refused_beyond_the_on_chip_sizes(sfb):
    P, q, A, l, u = sfb.random_qp_batch(1, 2, 100, 40, 0.5)
    with pytest.raises(sfb._capi.SfbError) as ei:
        sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=50), trace_rows=4)
    assert "128" in str(ei.value)


def test_verbose_single_dense_problem_table_is_the_solves_own(sfb, oracle, capfd):
    """verbose on ONE dense problem with n + m <= 128 prints the native table: its rows are the oracle's trace of the same
    solve (to the printed digits), no diagnostic-solve note."""
    P, q, A, l, u = sfb.random_qp_batch(4, 1, 40, 20, 1.0)
    prm = sfb.QPSolverParams(max_iter=3000, verbose=True)
    capfd.readouterr()
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    out = capfd.readouterr().out
    assert "Solving dense QP with n=20, m=40" in out and "diagnostic solve" not in out
    lines = [ln for ln in out.splitlines() if ":" in ln and ln.split(":")[0].strip().isdigit()]
    ref = oracle.qp_dense_solve_batch(P, q, A, l, u, params=_oracle_params(oracle, sfb.QPSolverParams(max_iter=3000)), trace_rows=len(lines) + 2)
    used = ref["trace"][0][ref["trace"][0][:, 0] >= 0]
    assert len(lines) == len(used) and int(r.iter[0]) == int(ref["iter"][0])
    for ln, row in zip(lines, used):
        it, rest = ln.split(":")
        vals = [float(v) for v in rest.split()[:3]]
        assert int(it) == int(row[0])
        for v, e in zip(vals, row[1:4]):
            assert v == float("%.6e" % e)


@pytest.mark.parametrize("n,m,B", [(10, 20, 96), (20, 40, 64), (40, 60, 32)])
def test_dense_phase_times_as_data(sfb, n, m, B):
    """qp_solver.hpp:550-565 as data for dense problems with n + m <= 128 (TRACE instance of the on-chip kernel): same results
    as the plain call, six non-negative times per problem; with the table, their first four reach past its last TIME stamp."""
    P, q, A, l, u = sfb.random_qp_batch(8, B, m, n, 1.0)
    prm = sfb.QPSolverParams(max_iter=1000)
    plain = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm, trace_rows=48, phases=True)
    assert np.array_equal(r.code, plain.code) and np.array_equal(r.iter, plain.iter)
    assert np.array_equal(r.primal, plain.primal, equal_nan=True) and np.array_equal(r.dual, plain.dual, equal_nan=True)
    ph = r.phase_us
    assert ph.shape == (B, 6) and (ph >= 0).all() and (ph[:, 1:4] > 0).all()
    used = r.trace[:, :, 0] >= 0
    last_time = np.where(used, r.trace[:, :, 4], 0.0).max(1)
    # (the table's clock starts at the reference's t0, after the scaling: phases 1 .. 3 cover it)
    assert (ph[:, 1:4].sum(1) + 0.02 >= last_time).all(), (ph[:4], last_time[:4])
    only = sfb.solve_qp_batch_host(P, q, A, l, u, prm, phases=True)
    assert only.trace is None and np.array_equal(only.primal, plain.primal, equal_nan=True) and (only.phase_us >= 0).all()


def test_verbose_single_dense_problem_prints_the_reference_summary(sfb, capfd):
    """... and the closing summary (:550-565) after the native table, its phase lines adding up to the total."""
    P, q, A, l, u = sfb.random_qp_batch(4, 1, 40, 20, 1.0)
    capfd.readouterr()
    r = sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=3000, verbose=True))
    out = capfd.readouterr().out
    tail = out[out.index("QP solver summary:"):]
    assert "Result %d" % int(r.code[0]) in tail and "NOTE the table's solve" not in out
    vals = {ln[:25].strip(): float(ln[25:]) for ln in tail.splitlines() if ln.startswith("  ")}
    assert set(vals) == {"Matrix filling", "Factorization", "Iteration", "Polish"}
    total = [float(ln.split()[-1]) for ln in tail.splitlines() if ln.startswith("Total time")][0]
    assert abs(sum(vals.values()) - total) <= 2.0
    assert int([ln for ln in tail.splitlines() if ln.startswith("Iterations")][0].split()[-1]) == int(r.iter[0]) - 1
