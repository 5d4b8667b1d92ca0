"""Parity of the HIP sparse (shared-pattern) QP kernel with the sparse CPU oracle.  Needs an MI355X.
Bar: codes and iteration counts bit-exact, primal/dual within 1e-8 (relative to 1+|value|)."""
import numpy as np
import pytest
import scipy.sparse as sp

from qp_cases import KNOWN_ANSWERS, is_approx
from sparse_cases import dense_batch_to_sparse
from test_qp_dense_gpu import _compare, _oracle_params

pytestmark = pytest.mark.gpu


from test_mpc_gpu import sweep_mode  # noqa: E402,F401  (fixture: plain / masked factor stream)


@pytest.mark.parametrize("name", sorted(KNOWN_ANSWERS))
def test_sparse_known_answers(sfb, oracle, name):
    case = KNOWN_ANSWERS[name]
    P, q, A, l, u = (np.asarray(t, dtype=np.float64) for t in case[:5])
    Pc = sp.csc_matrix(P); Pc.eliminate_zeros(); Pc.sort_indices()
    Ac = sp.csr_matrix(A); Ac.sort_indices()
    plan = sfb.SparseQPPlan(len(q), len(l), Pc.indptr, Pc.indices, Ac.indptr, Ac.indices)
    prm = sfb.QPSolverParams(max_iter=100000)
    r = plan.solve_batch_host(Pc.data[None], q[None], Ac.data[None], l[None], u[None], prm)
    code, primal, ptol, objv, otol = case[5:]
    assert int(r.code[0]) == code
    if primal is not None:
        assert is_approx(r.primal[0], primal, ptol)
    if objv is not None:
        assert abs(r.objective[0] - objv) <= otol
    ref = oracle.qp_sparse_solve_batch(Pc.indptr, Pc.indices, Pc.data[None], q[None], Ac.indptr, Ac.indices,
                                       Ac.data[None], l[None], u[None], perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm))
    _compare(r, ref)
    sol = sfb.solve_qp_sparse(sfb.QuadraticProgramSparse(P=Pc, q=q, A=Ac, l=l, u=u))
    assert int(sol.code) == code


@pytest.mark.parametrize("n,m,density,ordering", [(10, 20, 0.3, 1), (10, 20, 1.0, 0), (6, 9, 0.5, 1), (30, 50, 0.15, 1),
                                                  (80, 120, 0.05, 1)])
def test_random_sparse_batches(sfb, oracle, n, m, density, ordering, sweep_mode):
    B = 192
    P, q, A, l, u = sfb.random_qp_batch(13, B, m, n, density)
    rng = np.random.default_rng(n + m)
    l = np.where(rng.random((B, m)) < 0.3, u - 2 * rng.random((B, m)), l)
    l = np.where(rng.random((B, m)) < 0.1, u, l)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=(n % 2 == 0))
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, ordering=ordering)
    for prm in (sfb.QPSolverParams(max_iter=2000), sfb.QPSolverParams(max_iter=600, scaling=False, polish=False)):
        r = plan.solve_batch_host(Px, q, Ax, l, u, prm)
        ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                           params=_oracle_params(oracle, prm), nthreads=8)
        bit = _compare(r, ref)
        print(n, m, "nnzL", plan.nnzL, "bit-identical:", bit, "codes", np.bincount(r.code, minlength=7))
    # warm start
    prm = sfb.QPSolverParams(max_iter=2000)
    ok = np.isfinite(ref["x"]).all(1) & np.isfinite(ref["y"]).all(1)
    wx, wy = np.where(ok[:, None], ref["x"], 0.0), np.where(ok[:, None], ref["y"], 0.0)
    r2 = plan.solve_batch_host(Px, q + 0.01, Ax, l, u, prm, warm_x=wx, warm_y=wy)
    ref2 = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q + 0.01, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                        params=_oracle_params(oracle, prm), warm_x=wx, warm_y=wy, nthreads=8)
    _compare(r2, ref2)


@pytest.mark.parametrize("n,m,sci", [(10, 20, 2), (10, 20, 5), (24, 60, 3)])
def test_infeasibility_verdicts_with_a_check_every_few_iterations(sfb, oracle, n, m, sci):
    """The primal-infeasibility test decides from a bound on the ordered certificate sum wherever the bound allows
    (sp_check_stopping, FAST PATH; tests/test_check_bound.py): many checks per solve on batches in which every verdict occurs,
    codes / iteration counts / iterates identical to the oracle, which always forms the ordered sum."""
    B = 256
    P, q, A, l, u = sfb.random_qp_batch(20261301 + n, B, m, n, 0.5)
    rng = np.random.default_rng(n * m + sci)
    mask = rng.random((B, m))
    l = np.where(mask < 0.15, -np.inf, l); u = np.where((mask > 0.15) & (mask < 0.3), np.inf, u)
    l = np.where(mask > 0.85, u, l)              # equality rows (dense: mostly infeasible systems when m > n)
    P[B // 2:] *= 1e-3                           # weakly convex half: dual-infeasibility candidates with free rows
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=True)
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj)
    for prm in (sfb.QPSolverParams(max_iter=1500, stop_check_iter=sci), sfb.QPSolverParams(max_iter=400, stop_check_iter=sci, scaling=False, eps_abs=1e-6, eps_rel=1e-6)):
        r = plan.solve_batch_host(Px, q, Ax, l, u, prm)
        ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                           params=_oracle_params(oracle, prm), nthreads=8)
        _compare(r, ref)
        codes = np.bincount(r.code, minlength=7)
        print(n, m, sci, "codes", codes)
    assert codes[2] > 0  # SFB_QP_PRIMAL_INFEASIBLE verdicts (by certificate) occurred


def test_large_banded_problem_uses_element_indices(sfb, oracle, sweep_mode):
    """n + m = 9 000 > 8 190: the sweep schedules store element indices instead of 16-bit byte offsets and the
    work vector takes 72 KB of LDS per wave.  Tridiagonal P, three entries per row of A; bit-identical to the oracle."""
    import scipy.sparse as sp
    n, m, B = 5000, 4000, 3
    rng = np.random.default_rng(1)
    Pm = sp.diags([np.full(n, 2.0), np.full(n - 1, -0.5)], [0, 1], format="csc"); Pm.sort_indices()
    rows = np.repeat(np.arange(m), 3); cols = (np.arange(m)[:, None] + np.arange(3)[None, :]).ravel() % n
    Am = sp.csr_matrix((np.ones(3 * m), (rows, cols)), shape=(m, n)); Am.sort_indices()
    plan = sfb.SparseQPPlan(n, m, Pm.indptr, Pm.indices, Am.indptr, Am.indices)
    Px = np.tile(Pm.data, (B, 1)) * (1 + 0.1 * rng.random((B, Pm.nnz)))
    Ax = rng.uniform(-1, 1, (B, Am.nnz))
    q = rng.uniform(-1, 1, (B, n)); u = rng.uniform(0.5, 1.5, (B, m))
    l = np.where(rng.random((B, m)) < 0.5, -np.inf, u - 1.0)
    prm = sfb.QPSolverParams(max_iter=300)
    r = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    ref = oracle.qp_sparse_solve_batch(Pm.indptr.astype(np.int32), Pm.indices.astype(np.int32), Px, q,
                                       Am.indptr.astype(np.int32), Am.indices.astype(np.int32), Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm), nthreads=3)
    assert _compare(r, ref)


def test_non_finite_inputs_follow_the_reference_semantics_sparse(sfb, oracle):
    """NaN / inf in the values of a shared-pattern batch: no input validation in the reference -- the values propagate
    through the IEEE arithmetic; l = +inf / u = -inf is the pre-check's PrimalInfeasible.  Codes, iteration counts and
    the bit patterns of primal / dual (up to NaN payloads) equal the sparse oracle's; clean items are unaffected."""
    n, m, B = 30, 50, 12
    P, q, A, l, u = sfb.random_qp_batch(7, B, m, n, 0.15)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m)
    clean = (Px.copy(), q.copy(), Ax.copy(), l.copy(), u.copy())
    q[1, 0] = np.nan
    Px[2, 0] = np.nan
    Ax[3, 1] = np.inf
    l[4, 2] = np.inf
    u[5, 3] = -np.inf
    l[6, 0] = np.nan
    q[7, :] = np.inf
    Ax[8, :] = 0.0
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj)
    prm = sfb.QPSolverParams(max_iter=300)
    r = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm))
    assert np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"]), (r.code, ref["code"], r.iter, ref["iter"])
    assert np.array_equal(r.primal, ref["x"], equal_nan=True) and np.array_equal(r.dual, ref["y"], equal_nan=True)
    assert r.code[4] == 2 and r.code[5] == 2 and r.iter[4] == 0
    rc = plan.solve_batch_host(*clean, prm)
    for b in (0, 9, 10, 11):
        assert np.array_equal(rc.primal[b], r.primal[b]) and rc.iter[b] == r.iter[b]


@pytest.mark.parametrize("n,m,density,seed", [(10, 20, 0.5, 11), (30, 50, 0.15, 12), (16, 12, 0.4, 13)])
def test_non_finite_matrix_entries_at_random_places(sfb, oracle, n, m, density, seed):
    """inf / -inf / NaN at random positions of the STORED values of P and A (one to three per item, every item): whatever the
    reference's arithmetic makes of them -- including a solve that still ends Optimal and is polished, where the first refinement
    round of polish takes its residual as h (x = 0: csrc/qp_sparse.hip sp_polish; the reference forms h - H 0) -- the kernel makes
    the same of them: codes, iteration counts, and primal / dual bit for bit up to NaN payloads against the sparse oracle."""
    B = 24
    P, q, A, l, u = sfb.random_qp_batch(seed, B, m, n, density)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m)
    rng = np.random.default_rng(seed)
    vals = (np.inf, -np.inf, np.nan)
    for b in range(B):
        for _ in range(1 + rng.integers(3)):
            arr = Px if rng.random() < 0.4 and Px.shape[1] > 0 else Ax
            arr[b, rng.integers(arr.shape[1])] = vals[rng.integers(3)]
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj)
    prm = sfb.QPSolverParams(max_iter=200)
    r = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm))
    assert np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"]), (r.code, ref["code"], r.iter, ref["iter"])
    assert np.array_equal(r.primal, ref["x"], equal_nan=True) and np.array_equal(r.dual, ref["y"], equal_nan=True)
    assert np.array_equal(r.objective, ref["obj"], equal_nan=True)


def test_solve_after_solve_batch_does_not_reuse_a_foreign_factor(sfb):
    """QPSolver<QuadraticProgramSparse>::solve() flags reuse_factor when it is handed the previous solve()'s matrices
    again; solve_batch() on the same solver shares the device workspace, so it must end that claim (round-2 advisor
    finding): solve(A1), solve_batch(A2), solve(A1) -- scaling off, so the kernel's own re-check of c cannot tell the two
    problems apart -- gives the first call's bits again and is not flagged; one more solve(A1) is flagged, same bits."""
    import ctypes as C
    from examples import models_lib as M
    out = np.full(8, -1.0)
    assert M.lib().sfbx_test_solve_after_solve_batch(out.ctypes.data_as(C.c_void_p)) == 0
    assert out[0] == 1.0 and out[1] == 0.0 and out[2] == 1.0 and out[3] == 1.0, out
    assert not np.allclose(out[4:6], out[6:8])          # the batch call really solved another problem


@pytest.mark.parametrize("n,m,density", [(10, 20, 0.5), (30, 50, 0.15)])
def test_verbose_table_as_data_matches_the_oracle_trace(sfb, oracle, n, m, density):
    """qp_solver.hpp:490-501: ITER, OBJ, PRI_RES, DUA_RES per stopping check -- the device trace against the oracle's,
    and the results of the traced call against the plain one (bit for bit)."""
    B, rows = 96, 12
    P, q, A, l, u = sfb.random_qp_batch(17, B, m, n, density)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=True)
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj)
    prm = sfb.QPSolverParams(max_iter=252)
    plain = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    r = plan.solve_batch_host(Px, q, Ax, l, u, prm, trace_rows=rows)
    assert np.array_equal(r.code, plain.code) and np.array_equal(r.iter, plain.iter)
    assert np.array_equal(r.primal, plain.primal, equal_nan=True) and np.array_equal(r.dual, plain.dual, equal_nan=True)
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm), nthreads=8, trace_rows=rows)
    assert np.array_equal(r.iter, ref["iter"])
    tr, tref = r.trace, ref["trace"]
    assert np.array_equal(tr[:, :, 0], tref[:, :, 0]), "check iterations differ"
    used = tref[:, :, 0] >= 0
    assert used.sum() > B  # more than one check per item on average
    assert (tr[:, :, 0][used] % 25 == 1).all()  # checks at iter % 25 == 1 (:479)
    for col, name in ((1, "OBJ"), (2, "PRI_RES"), (3, "DUA_RES")):
        a, b = tr[:, :, col][used], tref[:, :, col][used]
        fin = np.isfinite(b)
        assert np.array_equal(a[fin], b[fin]), (name, np.abs(a[fin] - b[fin]).max())
    both = used[:, 1:] & used[:, :-1]
    assert (tr[:, :, 4][used] >= 0).all() and (tr[:, 1:, 4][both] >= tr[:, :-1, 4][both]).all()  # TIME (device clock, us) grows
    # a table shorter than the number of checks: the leading rows, nothing written behind them
    short = plan.solve_batch_host(Px, q, Ax, l, u, prm, trace_rows=2)
    assert short.trace.shape == (B, 2, 5) and np.array_equal(short.trace[:, :, :4], tr[:, :2, :4], equal_nan=True)


def test_verbose_single_problem_prints_the_reference_table(sfb, capfd):
    """QPSolverParams::verbose on one sparse problem: the header and one line per stopping check (qp_solver.hpp:409-420,
    :490-501), then the summary."""
    case = KNOWN_ANSWERS["PortfolioOptimization"] if "PortfolioOptimization" in KNOWN_ANSWERS else KNOWN_ANSWERS[sorted(KNOWN_ANSWERS)[0]]
    P, q, A, l, u = (np.asarray(t, dtype=np.float64) for t in case[:5])
    Pc = sp.csc_matrix(P); Pc.eliminate_zeros(); Pc.sort_indices()
    Ac = sp.csr_matrix(A); Ac.sort_indices()
    plan = sfb.SparseQPPlan(len(q), len(l), Pc.indptr, Pc.indices, Ac.indptr, Ac.indices)
    r = plan.solve_batch_host(Pc.data[None], q[None], Ac.data[None], l[None], u[None], sfb.QPSolverParams(verbose=True))
    out = capfd.readouterr().out
    assert "========================= QP Solver" in out and "Solving sparse QP with n=%d, m=%d" % (len(q), len(l)) in out
    assert "ITER" in out and "PRI_RES" in out and "DUA_RES" in out
    lines = [ln for ln in out.splitlines() if ln.strip().split(":")[0].strip().isdigit() and ":" in ln]
    assert len(lines) == (int(r.iter[0]) - 2) // 25 + 1 and lines[0].strip().startswith("1:")


def test_phase_times_as_data(sfb, oracle):
    """qp_solver.hpp:550-565 (Matrix filling / Factorization / Iteration / Polish) as data: the *_phases entry point solves
    through the TRACE instance -- same results as the plain call -- and returns six non-negative per-phase times per item whose
    sum covers the table's last TIME stamp (taken inside the iteration phase); for ONE problem alone on the device the sum is
    the kernel's duration as HIP events see it (2 % + launch latency)."""
    import torch
    B, n, m, rows = 64, 30, 50, 12
    P, q, A, l, u = sfb.random_qp_batch(23, B, m, n, 0.15)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=True)
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj)
    prm = sfb.QPSolverParams(max_iter=252)
    plain = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    r = plan.solve_batch_host(Px, q, Ax, l, u, prm, trace_rows=rows, phases=True)
    assert np.array_equal(r.code, plain.code) and np.array_equal(r.iter, plain.iter)
    assert np.array_equal(r.primal, plain.primal, equal_nan=True) and np.array_equal(r.dual, plain.dual, equal_nan=True)
    ph = r.phase_us
    assert ph.shape == (B, 6) and (ph >= 0).all() and (ph.sum(1) > 0).all()
    used = r.trace[:, :, 0] >= 0
    last_time = np.where(used, r.trace[:, :, 4], 0.0).max(1)
    assert (ph[:, :4].sum(1) + 0.02 >= last_time).all()  # scaling + fill + factor + iteration reach past the last check
    assert (ph[:, 4][plain.code != 0] < 5.0).all()       # no polish unless Optimal (a few clock reads only)
    only = plan.solve_batch_host(Px, q, Ax, l, u, prm, phases=True)  # without the table
    assert only.trace is None and np.array_equal(only.primal, plain.primal, equal_nan=True) and only.phase_us.shape == (B, 6)
    # one MPC-sized problem alone on the device: phase sum against HIP events around the launch
    from examples import models_lib as M
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(12, 50)
    Av, l, u = M.mpc_assemble_batch(12, 50, 1, seed=5, threads=1)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(12, 50), keep=(Av[0] != 0.0))
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (Pv[None], np.zeros((1, d["n"])), Av, l, u)]
    f64 = dict(dtype=torch.float64, device=dev)
    x, y = torch.empty((1, d["n"]), **f64), torch.empty((1, d["m"]), **f64)
    out = torch.empty((2, 1), dtype=torch.int32, device=dev)
    ws = torch.empty((plan.workspace_bytes(1) + 7) // 8, **f64)
    phd = torch.zeros((1, 6), **f64)
    st = torch.cuda.Stream()
    best = None
    with torch.cuda.stream(st):
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            plan.solve_batch_device_phases(1, *(a.data_ptr() for a in t), x.data_ptr(), y.data_ptr(), 0, out[0].data_ptr(),
                                           out[1].data_ptr(), ws.data_ptr(), phd.data_ptr(), sfb.QPSolverParams(), stream=st.cuda_stream)
            e1.record(st)
            st.synchronize()
            ev_us, sum_us = 1e3 * e0.elapsed_time(e1), float(phd.sum())
            if best is None or ev_us < best[0]:
                best = (ev_us, sum_us)
    ev_us, sum_us = best
    print("one MPC QP: events %.1f us, phases %s us" % (ev_us, np.round(phd.cpu().numpy()[0], 1)))
    assert int(out[1, 0]) == 0 and sum_us <= ev_us + 1.0 and sum_us >= 0.98 * ev_us - 40.0, (ev_us, sum_us)


def test_verbose_single_problem_prints_the_reference_summary(sfb, capfd):
    """... and the closing summary of qp_solver.hpp:550-565 with its four phase lines."""
    case = KNOWN_ANSWERS[sorted(KNOWN_ANSWERS)[0]]
    P, q, A, l, u = (np.asarray(t, dtype=np.float64) for t in case[:5])
    Pc = sp.csc_matrix(P); Pc.eliminate_zeros(); Pc.sort_indices()
    Ac = sp.csr_matrix(A); Ac.sort_indices()
    plan = sfb.SparseQPPlan(len(q), len(l), Pc.indptr, Pc.indices, Ac.indptr, Ac.indices)
    r = plan.solve_batch_host(Pc.data[None], q[None], Ac.data[None], l[None], u[None], sfb.QPSolverParams(verbose=True))
    out = capfd.readouterr().out
    tail = out[out.index("QP solver summary:"):]
    assert "Result %d" % int(r.code[0]) in tail
    for name in ("Iterations", "Total time", "  Matrix filling", "  Factorization", "  Iteration", "  Polish"):
        assert name in tail, name
    vals = {ln[:25].strip(): float(ln[25:]) for ln in tail.splitlines() if ln.startswith("  ")}
    total = [float(ln.split()[-1]) for ln in tail.splitlines() if ln.startswith("Total time")][0]
    assert abs(sum(vals.values()) - total) <= 2.0  # the four lines add up to the total (printed as whole microseconds)
    assert int([ln for ln in tail.splitlines() if ln.startswith("Iterations")][0].split()[-1]) == int(r.iter[0]) - 1


@pytest.mark.parametrize("units", [1, 0])
def test_zero_pivot_ends_with_unknown_in_both_factorisation_engines(sfb, oracle, knobs, units):
    """SimplicialLDLT info() == NumericalIssue -> QPSolutionStatus::Unknown (qp_solver.hpp:430-433): sigma = 0 and a zero on
    the diagonal of P give the KKT matrix a zero pivot for some items of the batch.  The unit engine of the numeric
    factorisation notices it when a segment's diagonal is final (after divisions by it have produced inf / NaN elsewhere),
    the supernodal engine (SFB_PLAN_UNITS=0) at the column: same verdicts, and the other items are untouched."""
    B, n, m = 48, 12, 18
    P, q, A, l, u = sfb.random_qp_batch(31, B, m, n, 0.4)
    Pm = P.reshape(B, n, n).copy()
    bad = np.arange(B) % 3 == 0
    Pm[bad, :, 0] = 0.0
    Pm[bad, 0, :] = 0.0                                      # first variable: no quadratic cost at all
    P = Pm.reshape(B, n * n)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=True)
    if not units:
        knobs.set(SFB_PLAN_UNITS=0)
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, ordering=0)   # natural order: the zero comes first
    prm = sfb.QPSolverParams(sigma=0.0, max_iter=400, scaling=False)
    r = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm), nthreads=4)
    assert np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"])
    assert (r.code[bad] == 6).all() and (r.iter[bad] == 0).all() and (r.code[~bad] != 6).any()
    good = r.code != 6
    assert np.array_equal(r.primal[good], ref["x"][good], equal_nan=True) and np.array_equal(r.dual[good], ref["y"][good], equal_nan=True)


def test_host_entry_writes_into_the_callers_result_buffers(sfb):
    """solve_batch_host(..., out=earlier solution): the C entry writes into the caller's x / y / obj / iter / code, so a control
    loop keeps its result arrays from tick to tick.  Same bits as a call that allocates; the arrays are the very ones handed
    in; arrays of another batch size or dtype are refused before anything is launched."""
    n, m, B = 10, 20, 96
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 0.4)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=True)
    plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj)
    prm = sfb.QPSolverParams(max_iter=2000)
    fresh = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    keep = plan.solve_batch_host(Px, q + 1.0, Ax, l, u, prm)       # (other contents, to be overwritten)
    ids = [id(a) for a in (keep.primal, keep.dual, keep.objective, keep.iter, keep.code)]
    again = plan.solve_batch_host(Px, q, Ax, l, u, prm, out=keep)
    assert [id(a) for a in (again.primal, again.dual, again.objective, again.iter, again.code)] == ids
    for a, b in ((again.primal, fresh.primal), (again.dual, fresh.dual), (again.objective, fresh.objective),
                 (again.iter, fresh.iter), (again.code, fresh.code)):
        assert np.array_equal(a, b, equal_nan=True)
    with pytest.raises(ValueError):
        plan.solve_batch_host(Px[:8], q[:8], Ax[:8], l[:8], u[:8], prm, out=keep)
    bad = plan.solve_batch_host(Px, q, Ax, l, u, prm)
    bad.iter = bad.iter.astype(np.int64)
    with pytest.raises(ValueError):
        plan.solve_batch_host(Px, q, Ax, l, u, prm, out=bad)
    plan.close()
