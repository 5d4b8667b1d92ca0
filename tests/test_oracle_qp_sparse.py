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


def _compress(Ap, Aj, keep):
    rows = np.repeat(np.arange(len(Ap) - 1), np.diff(Ap))
    cnt = np.zeros(len(Ap), np.int64)
    np.add.at(cnt, rows[keep] + 1, 1)
    return np.cumsum(cnt).astype(np.int32), np.ascontiguousarray(Aj[keep])


@pytest.mark.parametrize("variant,K,B", [(6, 10, 12), (12, 50, 6)])
def test_explicit_zeros_can_be_left_out_of_the_analysis(oracle, sfb, variant, K, B):
    """The exactness argument behind sfb_sparse_qp_plan_create_pruned, checked on the CPU: ocp_to_qp stores dense
    Jacobian blocks (block_add, utils/sparse.hpp:33-50; ocp_to_qp.hpp:258-264), most of whose entries are 0.0 in
    every agent.  Solving on the kept entries only gives the results of the whole stored pattern under the same
    elimination order: codes, iterations, primal, dual, objective all compare equal (zeros may differ in sign)."""
    from examples import models_lib as M
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3)
    keep = np.any(Av != 0.0, axis=0)
    assert keep.sum() < 0.5 * keep.size           # the headline pattern: 4 076 of 12 832 stored entries
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    whole = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    assert plan.pruned and not whole.pruned
    assert plan.nnzA == keep.size and plan.nnzA_kept == int(keep.sum())
    assert plan.nnzL < 0.5 * whole.nnzL
    if (variant, K) == (12, 50):
        assert (whole.nnzL, plan.nnzL) == (42182, 14862)   # separators in nested-dissection order (MPC front)
    # workspace: exact requirement of a call vs the per-item bound
    assert plan.workspace_bytes(1) <= plan.workspace_bytes_per_item
    assert plan.workspace_bytes(4096) <= 4096 * plan.workspace_bytes_per_item
    assert plan.workspace_bytes(4096) < whole.workspace_bytes(4096)
    Ap2, Aj2 = _compress(Ap, Aj, keep)
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    prm = oracle.default_params(max_iter=4000)
    # (the summation order of the factorisation is a postorder of the elimination tree, which differs between the
    #  two patterns: the whole-pattern run is told to use the pruned one's; the pruned run finds it by itself)
    full = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(), params=prm, nthreads=4)
    comp = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap2, Aj2, np.ascontiguousarray(Av[:, keep]), l, u,
                                        perm=plan.perm, params=prm, nthreads=4)
    comp2 = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap2, Aj2, np.ascontiguousarray(Av[:, keep]), l, u,
                                         perm=plan.perm, forder=plan.factor_order(), params=prm, nthreads=4)
    assert np.array_equal(comp["x"], comp2["x"]) and np.array_equal(comp["iter"], comp2["iter"])
    fo = plan.factor_order()
    assert sorted(fo.tolist()) == list(range(d["n"] + d["m"]))
    assert full["nnzL"] == plan.nnzL_fallback and comp["nnzL"] == plan.nnzL
    for key in ("code", "iter", "x", "y", "obj"):
        assert np.array_equal(full[key], comp[key]), key
    assert (full["code"] == 0).all()


def test_pruned_plan_argument_checks(sfb):
    Pp, Pi, Ap, Aj = [0, 1, 2], [0, 1], [0, 2, 4], [0, 1, 0, 1]
    with pytest.raises(ValueError):
        sfb.SparseQPPlan(2, 2, Pp, Pi, Ap, Aj, keep=[1, 0, 1])           # wrong length
    p = sfb.SparseQPPlan(2, 2, Pp, Pi, Ap, Aj, keep=[1, 1, 1, 1])           # nothing masked: a plain plan
    assert not p.pruned and p.nnzA_kept == 4
    p = sfb.SparseQPPlan(2, 2, Pp, Pi, Ap, Aj, keep=[1, 0, 0, 1])
    assert p.pruned and p.nnzA_kept == 2 and p.nnzL <= p.nnzL_fallback
    p = sfb.SparseQPPlan(2, 2, Pp, Pi, Ap, Aj, keep=[0, 0, 0, 0])           # every stored entry declared zero
    assert p.pruned and p.nnzA_kept == 0
