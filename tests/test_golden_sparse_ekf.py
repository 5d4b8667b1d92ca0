"""Committed golden vectors of the sparse MPC path and of the EKF path (tests/golden/mpc_sparse.npz, ekf.npz,
written by tests/golden/make_golden_sparse.py).  CPU: the oracle still reproduces them (regression guard of the
checker itself, and of the host front whose pattern / elimination order they record).  GPU: the HIP path through
the C-ABI reproduces them bit for bit."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mpc():
    g = np.load(os.path.join(GOLD, "mpc_sparse.npz"))
    B = g["l"].shape[0]
    Av = np.zeros((B, g["keep"].size))
    Av[:, g["keep"]] = g["Av_kept"]
    return g, Av, np.tile(g["Pv"], (B, 1)), np.zeros((B, int(g["n"])))


def test_mpc_fixture_pattern_is_what_the_front_produces(sfb):
    """The fixture records the transcription's pattern, the elimination stages and the plan's orders: the C++ MPC
    front and the planner still produce exactly these (a change of either shows up here, on the CPU)."""
    from examples import models_lib as M
    g, Av, Px, q = _mpc()
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(int(g["variant"]), int(g["K"]))
    for a, b in ((Pp, g["Pp"]), (Pi, g["Pi"]), (Pv, g["Pv"]), (Ap, g["Ap"]), (Aj, g["Aj"]), (M.mpc_stage(12, 50), g["stage"])):
        assert np.array_equal(a, b)
    Av2, l2, u2 = M.mpc_assemble_batch(12, 50, Av.shape[0], seed=3)
    assert np.array_equal(Av2, Av) and np.array_equal(l2, g["l"]) and np.array_equal(u2, g["u"])
    plan = sfb.SparseQPPlan(int(g["n"]), int(g["m"]), Pp, Pi, Ap, Aj, stage=g["stage"], keep=g["keep"])
    assert np.array_equal(plan.perm, g["perm"]) and np.array_equal(plan.factor_order(), g["forder"])


def test_oracle_reproduces_mpc_fixture(oracle):
    g, Av, Px, q = _mpc()
    prm = oracle.default_params()
    r = oracle.qp_sparse_solve_batch(g["Pp"], g["Pi"], Px, q, g["Ap"], g["Aj"], Av, g["l"], g["u"], perm=g["perm"],
                                     forder=g["forder"], params=prm, nthreads=4)
    for k in ("code", "iter", "x", "y", "obj"):
        assert np.array_equal(r[k], g["cold_" + k]), k


def test_oracle_reproduces_ekf_fixture(oracle):
    g = np.load(os.path.join(GOLD, "ekf.npz"))
    for tag in ("se2r3", "generic"):
        dof, ny, B = [int(v) for v in g[tag + "_dims"]]
        P = g[tag + "_P0"]
        for tick in range(3):
            Pp = oracle.ekf_predict_batch(g[tag + "_A"], g[tag + "_Q"], g[tag + "_dt"], P)
            P, delta, _ = oracle.ekf_update_batch(g[tag + "_H"], g[tag + "_R"], g[tag + "_r"][tick], Pp, dof)
            assert np.array_equal(P, g[tag + "_P_ticks"][tick]) and np.array_equal(delta, g[tag + "_delta_ticks"][tick])
        assert np.array_equal(oracle.ekf_predict_batch(g[tag + "_A"], g[tag + "_Q"], g[tag + "_dt"], g[tag + "_P0"], stepper="rk4"),
                              g[tag + "_P_rk4"])


@pytest.mark.gpu
def test_hip_sparse_path_reproduces_mpc_fixture(sfb):
    g, Av, Px, q = _mpc()
    plan = sfb.SparseQPPlan(int(g["n"]), int(g["m"]), g["Pp"], g["Pi"], g["Ap"], g["Aj"], stage=g["stage"], keep=g["keep"])
    assert np.array_equal(plan.perm, g["perm"])
    r = plan.solve_batch_host(Px, q, Av, g["l"], g["u"], sfb.QPSolverParams())
    for got, k in ((r.code, "code"), (r.iter, "iter"), (r.primal, "x"), (r.dual, "y"), (r.objective, "obj")):
        assert np.array_equal(got, g["cold_" + k]), k
    l2, u2 = g["l"] + 1e-3 * (g["l"] == g["u"]), g["u"] + 1e-3 * (g["l"] == g["u"])
    r2 = plan.solve_batch_host(Px, q, Av, l2, u2, sfb.QPSolverParams(), warm_x=r.primal, warm_y=r.dual)
    for got, k in ((r2.code, "code"), (r2.iter, "iter"), (r2.primal, "x"), (r2.dual, "y")):
        assert np.array_equal(got, g["warm_" + k]), k
    # the plan WITHOUT the explicit-zero declaration sums in another order (other elimination tree): same codes and
    # iteration counts here, values to rounding
    whole = sfb.SparseQPPlan(int(g["n"]), int(g["m"]), g["Pp"], g["Pi"], g["Ap"], g["Aj"], user_perm=g["perm"])
    rw = whole.solve_batch_host(Px, q, Av, g["l"], g["u"], sfb.QPSolverParams())
    assert np.array_equal(rw.code, g["cold_code"]) and np.abs(rw.primal - g["cold_x"]).max() < 1e-6


@pytest.mark.gpu
def test_hip_ekf_path_reproduces_fixture(sfb):
    g = np.load(os.path.join(GOLD, "ekf.npz"))
    for tag in ("se2r3", "generic"):
        dof, ny, B = [int(v) for v in g[tag + "_dims"]]
        P = g[tag + "_P0"]
        for tick in range(3):
            P, delta, info = sfb.ekf_step_batch_host(P, dof, A=g[tag + "_A"], Q=g[tag + "_Q"], dt=g[tag + "_dt"], H=g[tag + "_H"],
                                                     R=g[tag + "_R"], r=g[tag + "_r"][tick])
            assert (info == 0).all()
            assert np.array_equal(P, g[tag + "_P_ticks"][tick]) and np.array_equal(delta, g[tag + "_delta_ticks"][tick])
        assert np.array_equal(sfb.ekf_predict_batch_host(g[tag + "_P0"], dof, g[tag + "_A"], g[tag + "_Q"], g[tag + "_dt"], stepper="rk4"),
                              g[tag + "_P_rk4"])
        assert np.array_equal(sfb.ekf_predict_batch_host(g[tag + "_P0"], dof, g[tag + "_A"], g[tag + "_Q"], g[tag + "_dt"], stepper="rk4",
                                                         A_mid=g[tag + "_A_mid"], A_end=g[tag + "_A_end"]), g[tag + "_P_rk4_tv"])
