"""Pins the CPU oracle (oracle/qp_oracle.c) against every known answer the reference's own
tests hold for the QP path (tests/test_qp.cpp:54-336).  CPU only."""
import json
import os

import numpy as np
import pytest

from qp_cases import KNOWN_ANSWERS, as_batch, is_approx

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", sorted(KNOWN_ANSWERS))
def test_known_answer(oracle, name):
    case = KNOWN_ANSWERS[name]
    P, q, A, l, u = as_batch(case)
    # test_prm of tests/test_qp.cpp:32-35: defaults with polish
    r = oracle.qp_dense_solve_batch(P, q, A, l, u, params=oracle.default_params())
    code, primal, ptol, objv, otol = case[5:]
    assert int(r["code"][0]) == code
    if primal is not None:
        assert is_approx(r["x"][0], primal, ptol), (r["x"][0], primal)
    if objv is not None:
        assert abs(r["obj"][0] - objv) <= otol
    # hot start from own solution (tests/test_qp.cpp:69-72 etc.)
    r2 = oracle.qp_dense_solve_batch(P, q, A, l, u, params=oracle.default_params(), warm_x=r["x"], warm_y=r["y"])
    assert int(r2["code"][0]) == code
    if primal is not None:
        assert is_approx(r2["x"][0], primal, ptol)


def test_precheck_iter_zero(oracle):
    """qp_solver.hpp:361-364: the trivial infeasibility pre-check exits with iter == 0."""
    P, q, A, l, u = as_batch(KNOWN_ANSWERS["PrimalInfeasibleEasy"])
    r = oracle.qp_dense_solve_batch(P, q, A, l, u)
    assert int(r["iter"][0]) == 0 and int(r["code"][0]) == 2


def test_iteration_counts_match_survey(oracle):
    """Iteration counts are checks at iter = 1, 26, 51, ... reported +1 (qp_solver.hpp:465,548)."""
    exp = {"Basic": 27, "Unconstrained": 27, "HalfConstrained": 27, "PrimalInfeasibleEasy": 0,
           "PrimalInfeasibleHard": 27, "PrimalInfeasibleInfinity": 27, "DualInfeasible": 2,
           "PortfolioOptimization": 152, "TwoDimensional": 27}
    for name, it in exp.items():
        r = oracle.qp_dense_solve_batch(*as_batch(KNOWN_ANSWERS[name]))
        assert int(r["iter"][0]) == it, name


def test_max_iter_and_stop_check_quirk(oracle):
    """stop_check_iter == 1: `iter % 1 == 1` is never true (qp_solver.hpp:465) -> only max_iter stops."""
    P, q, A, l, u = as_batch(KNOWN_ANSWERS["Basic"])
    r = oracle.qp_dense_solve_batch(P, q, A, l, u, params=oracle.default_params(stop_check_iter=1, max_iter=40))
    assert int(r["code"][0]) == 4 and int(r["iter"][0]) == 40


def test_ldlt_reconstruction(oracle):
    """P A P' = L D L' for the restated Eigen LDLT on a quasi-definite matrix, and solve residual."""
    import ctypes as C
    rng = np.random.default_rng(0)
    k = 12
    G = rng.standard_normal((k, k))
    S = G @ G.T + np.eye(k)
    S[8:, 8:] = -S[8:, 8:]           # indefinite (quasi-definite-like)
    S[8:, :8] = G[8:, :8]
    S[:8, 8:] = G[8:, :8].T
    W = np.tril(S).copy()
    tr = np.zeros(k, dtype=np.int32)
    dp = C.POINTER(C.c_double)
    ok = oracle.lib().oracle_ldlt_factor(k, W.ctypes.data_as(dp), k, tr.ctypes.data_as(C.POINTER(C.c_int)))
    assert ok == 1
    L = np.tril(W, -1) + np.eye(k)
    D = np.diag(np.diag(W))
    perm = np.arange(k)
    for i in range(k):
        perm[[i, tr[i]]] = perm[[tr[i], i]]
    Sp = S[np.ix_(perm, perm)]
    assert np.allclose(L @ D @ L.T, Sp, atol=1e-10)
    b = rng.standard_normal(k)
    x = b.copy()
    oracle.lib().oracle_ldlt_solve(k, W.ctypes.data_as(dp), k, tr.ctypes.data_as(C.POINTER(C.c_int)),
                                   x.ctypes.data_as(dp))
    assert np.allclose(S @ x, b, atol=1e-9)


def test_golden_fixture_matches_oracle(oracle):
    """The committed golden vectors (tests/golden/make_golden.py) are reproduced by the oracle."""
    path = os.path.join(GOLD, "qp_dense_random.npz")
    g = np.load(path)
    prm = json.loads(str(g["params_json"]))
    for tag in prm:
        p = oracle.default_params(**prm[tag])
        r = oracle.qp_dense_solve_batch(g["P"], g["q"], g["A"], g["l"], g["u"], params=p)
        assert np.array_equal(r["code"], g[tag + "_code"])
        assert np.array_equal(r["iter"], g[tag + "_iter"])
        assert np.array_equal(r["x"], g[tag + "_x"])
        assert np.array_equal(r["y"], g[tag + "_y"])
