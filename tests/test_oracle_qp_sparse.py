"""Pins the sparse CPU oracle (oracle/qp_sparse_oracle.c): sparse known answers of the reference's
tests/test_qp.cpp and dense == sparse agreement (TwoDimensional :314-336), plus the product's
symbolic analysis (host code, no GPU) against the oracle's own symbolic pass."""
import numpy as np
import pytest
import scipy.sparse as sp

from qp_cases import KNOWN_ANSWERS, is_approx
from sparse_cases import dense_batch_to_sparse


def _sparse_case(case):
    P, q, A, l, u = (np.asarray(t, dtype=np.float64) for t in case[:5])
    Pc = sp.csc_matrix(P); Pc.eliminate_zeros(); Pc.sort_indices()
    Ac = sp.csr_matrix(A); Ac.sort_indices()
    return Pc, q, Ac, l, u


@pytest.mark.parametrize("name", sorted(KNOWN_ANSWERS))
def test_sparse_known_answers_and_dense_agreement(oracle, name):
    """BasicSparse (:103-122), PortfolioOptimizationSparse (:277-312) and every other known answer
    solved through the sparse branch; codes/iterations equal the dense oracle's, primal isApprox."""
    case = KNOWN_ANSWERS[name]
    Pc, q, Ac, l, u = _sparse_case(case)
    r = oracle.qp_sparse_solve_batch(Pc.indptr, Pc.indices, Pc.data[None], q[None], Ac.indptr, Ac.indices,
                                     Ac.data[None], l[None], u[None])
    code, primal, ptol, objv, otol = case[5:]
    assert int(r["code"][0]) == code
    if primal is not None:
        assert is_approx(r["x"][0], primal, ptol)
    if objv is not None:
        assert abs(r["obj"][0] - objv) <= otol
    P, A = np.asarray(case[0], float), np.asarray(case[2], float)
    rd = oracle.qp_dense_solve_batch(P.flatten("F")[None], q[None], A.flatten("F")[None], l[None], u[None])
    assert int(rd["code"][0]) == int(r["code"][0]) and int(rd["iter"][0]) == int(r["iter"][0])
    if code == 0:  # tests/test_qp.cpp:332-333  sol.primal.isApprox(sp_sol.primal) (1e-12)
        assert is_approx(r["x"][0], rd["x"][0], 1e-10) and is_approx(r["y"][0] + 1.0, rd["y"][0] + 1.0, 1e-10)
    # hot start
    r2 = oracle.qp_sparse_solve_batch(Pc.indptr, Pc.indices, Pc.data[None], q[None], Ac.indptr, Ac.indices,
                                      Ac.data[None], l[None], u[None], warm_x=r["x"], warm_y=r["y"])
    assert int(r2["code"][0]) == code


def test_any_elimination_order_gives_the_same_answers(oracle, sfb):
    rng = np.random.default_rng(1)
    n, m, B = 9, 14, 16
    P, q, A, l, u = sfb.random_qp_batch(3, B, m, n, 0.4)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m)
    prm = oracle.default_params(max_iter=4000)
    rd = oracle.qp_dense_solve_batch(P, q, A, l, u, params=prm)
    for perm in (None, rng.permutation(n + m), np.arange(n + m)[::-1].copy()):
        r = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=perm, params=prm)
        assert np.array_equal(r["code"], rd["code"]) and np.array_equal(r["iter"], rd["iter"])
        ok = rd["code"] == 0
        assert np.abs(r["x"] - rd["x"])[ok].max() < 1e-6


def test_plan_symbolic_matches_oracle_symbolic(oracle, sfb):
    """Product's host-side symbolic analysis (sparse_plan.cpp) vs the oracle's own: same nnz(L) for
    the same elimination order; minimum degree never worse than natural on these patterns."""
    n, m, B = 10, 20, 4
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 0.3)
    Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=True)
    for ordering in (0, 1):
        plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, ordering=ordering)
        perm = plan.perm
        assert sorted(perm.tolist()) == list(range(n + m))
        r = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Ax, l, u, perm=perm,
                                         params=oracle.default_params(max_iter=50))
        assert r["nnzL"] == plan.nnzL
        assert plan.workspace_bytes_per_item > 16 * plan.nnzL
        plan.close()
    nat = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, ordering=0)
    md = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, ordering=1)
    assert md.nnzL <= nat.nnzL
    up = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, user_perm=np.arange(n + m)[::-1].copy())
    assert np.array_equal(up.perm, np.arange(n + m)[::-1])


def test_plan_rejects_malformed_patterns(sfb):
    with pytest.raises(sfb._capi.SfbError):
        sfb.SparseQPPlan(2, 2, [0, 1, 2], [0, 5], [0, 1, 2], [0, 1])          # row index out of range
    with pytest.raises(sfb._capi.SfbError):
        sfb.SparseQPPlan(2, 2, [0, 1, 2], [0, 1], [0, 2, 3], [1, 0, 1])       # unsorted row of A
    with pytest.raises(sfb._capi.SfbError):
        sfb.SparseQPPlan(2, 2, [0, 1, 2], [0, 1], [0, 1, 2], [0, 1], user_perm=[0, 0, 1, 2])
