"""MPC path on the GPU: QPs produced by the host-side MPC transcription (C++ front, examples/models.cpp)
solved by the sparse HIP kernel, checked against the sparse CPU oracle; closed-loop behaviour of
tests/test_mpc.cpp.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

from examples import models_lib as M
from test_qp_dense_gpu import _compare, _oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "masked", "plain"])
def sweep_mode(request, knobs):
    """The sparse kernel streams the factor with plain loads or, while many waves are resident, with loads masked
    to the lanes that carry entries (no HBM traffic for the padding of the sweep schedule).  'auto' is the
    product's choice (plain for these small batches); the other two force one form (debug knob SFB_SP_LEAN_WAVES, read at
    every launch).  All three must agree with the oracle bit for bit."""
    if request.param != "auto":
        knobs.set(SFB_SP_LEAN_WAVES="-1" if request.param == "masked" else "1000000000")
    yield request.param


@pytest.mark.parametrize("variant,K,batch", [(6, 10, 48), (6, 50, 32), (12, 50, 24)])
def test_mpc_qp_batch_matches_oracle(sfb, oracle, variant, K, batch, sweep_mode):
    """BASELINE configs[2] problem (variant 12, K=50: n = m = 740) at oracle-sized batches; default
    MPCParams.qp (eps 1e-3, scaling, polish), cold start and warm start from the previous solution."""
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, batch, seed=3)
    Px = np.tile(Pv, (batch, 1))
    q = np.zeros((batch, d["n"]))
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    prm = sfb.QPSolverParams(max_iter=4000)
    r = plan.solve_batch_host(Px, q, Av, l, u, prm)
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm), nthreads=8)
    bit = _compare(r, ref)
    assert (r.code == 0).all(), np.bincount(r.code, minlength=7)   # MPC problems are feasible: Optimal
    print("variant", variant, "K", K, "nnzL", plan.nnzL, "iters", np.unique(r.iter), "bit-identical", bit)
    # u_0 (the MPC output, mpc.hpp:518) agrees with the oracle: BASELINE metric 'max |du| vs CPU ref'
    ub = d["Nx"] * (d["N"] + 1)
    assert np.abs(r.primal[:, ub:ub + 2] - ref["x"][:, ub:ub + 2]).max() <= 1e-8
    # input constraint -0.5 <= u <= 0.5 holds on the whole horizon (cr rows), initial state is pinned (ce rows)
    uu = r.primal[:, ub:]
    assert uu.min() >= -0.5 - 1e-2 and uu.max() <= 0.5 + 1e-2   # to the solver tolerance (eps 1e-3)
    # warm start from the solution of a slightly different problem
    r2 = plan.solve_batch_host(Px, q, Av, l + 1e-3 * (l == u), u + 1e-3 * (l == u), prm, warm_x=r.primal, warm_y=r.dual)
    ref2 = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l + 1e-3 * (l == u), u + 1e-3 * (l == u),
                                        perm=plan.perm, forder=plan.factor_order(), params=_oracle_params(oracle, prm), warm_x=ref["x"],
                                        warm_y=ref["y"], nthreads=8)
    _compare(r2, ref2)
    assert r2.iter.mean() <= r.iter.mean()


@pytest.mark.parametrize("variant,K,batch", [(6, 10, 48), (6, 50, 32), (12, 50, 24)])
def test_mpc_pruned_plan_equals_whole_pattern_oracle(sfb, oracle, variant, K, batch, sweep_mode):
    """sfb_sparse_qp_plan_create_pruned: the kernel analyses only the stored entries of A that are non-zero in a
    sample of the batch (the rest are explicit zeros of ocp_to_qp's dense Jacobian blocks) and must reproduce the
    oracle run on the WHOLE stored pattern with the same elimination order: codes, iterations, primal, dual and
    objective compare equal (numpy equality: zeros may differ in sign).  Cold and warm start."""
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, batch, seed=3)
    Px = np.tile(Pv, (batch, 1))
    q = np.zeros((batch, d["n"]))
    keep = np.any(Av[:8] != 0.0, axis=0)          # probe a few agents; the device guard covers the rest
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    assert plan.pruned and plan.nnzL < plan.nnzL_fallback
    prm = sfb.QPSolverParams(max_iter=4000)
    op = _oracle_params(oracle, prm)
    r = plan.solve_batch_host(Px, q, Av, l, u, prm)
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(), params=op, nthreads=8)
    assert ref["nnzL"] == plan.nnzL_fallback
    for a, b in ((r.code, ref["code"]), (r.iter, ref["iter"]), (r.primal, ref["x"]), (r.dual, ref["y"]),
                 (r.objective, ref["obj"])):
        assert np.array_equal(a, b)
    assert (r.code == 0).all()
    l2, u2 = l + 1e-3 * (l == u), u + 1e-3 * (l == u)
    r2 = plan.solve_batch_host(Px, q, Av, l2, u2, prm, warm_x=r.primal, warm_y=r.dual)
    ref2 = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l2, u2, perm=plan.perm, forder=plan.factor_order(), params=op, warm_x=ref["x"],
                                        warm_y=ref["y"], nthreads=8)
    for a, b in ((r2.code, ref2["code"]), (r2.iter, ref2["iter"]), (r2.primal, ref2["x"]), (r2.dual, ref2["y"])):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("batch,nbad", [(1, 1), (40, 7), (40, 40), (300, 3), (200, 150)])
def test_pruned_plan_guard_falls_back_to_the_whole_pattern(sfb, oracle, batch, nbad):
    """Items whose masked entries are NOT all zero are detected on the device and solved on the whole pattern
    (same elimination order) inside the same launch, in a pool of 64 whole-pattern workspace slots; the other items
    of the batch are unaffected.  Every item equals the whole-pattern oracle.  Cases: one item, a few, all items
    bad, more bad items than pool slots (slots are reused as their items finish)."""
    variant, K = 6, 10
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, batch, seed=11)
    keep = np.any(Av != 0.0, axis=0)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    rng = np.random.default_rng(5)
    bad = rng.choice(batch, nbad, replace=False)
    masked = np.nonzero(~keep)[0]
    for b in bad:                                   # a few masked entries become non-zero (one of them NaN-free but tiny)
        for e in rng.choice(masked, 3, replace=False):
            Av[b, e] = rng.uniform(-0.3, 0.3)
    Px, q = np.tile(Pv, (batch, 1)), np.zeros((batch, d["n"]))
    prm = sfb.QPSolverParams(max_iter=4000)
    r = plan.solve_batch_host(Px, q, Av, l, u, prm)
    # the whole-pattern oracle with the plan's elimination order; the items on the fallback path sum the
    # contributions to an entry of L in the order of the whole-pattern analysis, the others in that of the pruned one
    isbad = np.zeros(batch, bool); isbad[bad] = True
    for sel, fo in ((~isbad, plan.factor_order()), (isbad, plan.factor_order(fallback=True))):
        if not sel.any():
            continue
        ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px[sel], q[sel], Ap, Aj, Av[sel], l[sel], u[sel], perm=plan.perm, forder=fo,
                                           params=_oracle_params(oracle, prm), nthreads=8)
        for a, b in ((r.code, ref["code"]), (r.iter, ref["iter"]), (r.primal, ref["x"]), (r.dual, ref["y"]),
                     (r.objective, ref["obj"])):
            assert np.array_equal(a[sel], b, equal_nan=True)
    # a plain plan of the whole pattern with the same elimination order IS the fallback analysis
    plain = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, user_perm=plan.perm)
    assert np.array_equal(plain.factor_order(), plan.factor_order(fallback=True))
    rp = plain.solve_batch_host(Px, q, Av, l, u, prm)
    assert np.array_equal(rp.primal[isbad], r.primal[isbad], equal_nan=True) and np.array_equal(rp.iter[isbad], r.iter[isbad])
    assert np.allclose(rp.primal[~isbad], r.primal[~isbad], rtol=0, atol=1e-6)


@pytest.mark.parametrize("grid,slice_iters", [(7, 25), (32, 50), (64, 1)])
def test_time_sliced_launch_equals_one_block_per_item(sfb, oracle, grid, slice_iters, knobs):
    """Batches larger than the chip holds run on a persistent grid: an item that has used its slice while others
    wait is suspended (its state stays in its workspace) and continued later by another block.  Forced here with a
    tiny grid and short slices (SFB_SP_GRID / SFB_SP_SLICE are test / tuning knobs read at every launch): the results
    must equal the one-block-per-item launch and the oracle bit for bit, for a plain and for a pruned plan, cold
    and warm start, and with some items on the fallback path."""
    variant, K, B = 6, 10, 150
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=21)
    keep = np.any(Av != 0.0, axis=0)
    Av[5, np.nonzero(~keep)[0][3]] = 0.125      # one item violates the mask: fallback launch next to the sliced one
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    prm = sfb.QPSolverParams(max_iter=4000)
    for kp in (None, keep):
        plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=kp)
        knobs.set(SFB_SP_SLICE="0")
        base = plan.solve_batch_host(Px, q, Av, l, u, prm)
        knobs.set(SFB_SP_SLICE=str(slice_iters))
        knobs.set(SFB_SP_GRID=str(grid))
        r = plan.solve_batch_host(Px, q, Av, l, u, prm)
        r2 = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.5 * r.primal, warm_y=0.5 * r.dual)
        knobs.clear("SFB_SP_GRID")
        knobs.set(SFB_SP_SLICE="0")
        base2 = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.5 * r.primal, warm_y=0.5 * r.dual)
        for a, b in ((r, base), (r2, base2)):
            assert np.array_equal(a.code, b.code) and np.array_equal(a.iter, b.iter)
            assert np.array_equal(a.primal, b.primal) and np.array_equal(a.dual, b.dual)
            assert np.array_equal(a.objective, b.objective)
        assert r.iter.max() > 3 * max(25, slice_iters)      # items were suspended several times
        ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(),
                                           params=_oracle_params(oracle, prm), nthreads=8)
        ok = np.ones(B, bool)
        ok[5] = kp is None                        # pruned plan: item 5 went through the fallback analysis
        assert np.array_equal(r.iter[ok], ref["iter"][ok]) and np.array_equal(r.primal[ok], ref["x"][ok])
        ref5 = oracle.qp_sparse_solve_batch(Pp, Pi, Px[5:6], q[5:6], Ap, Aj, Av[5:6], l[5:6], u[5:6], perm=plan.perm,
                                            forder=plan.factor_order(fallback=True), params=_oracle_params(oracle, prm))
        assert np.array_equal(r.iter[5:6], ref5["iter"]) and np.array_equal(r.primal[5:6], ref5["x"])


@pytest.mark.parametrize("grid,slice_iters", [(16, 25), (48, 50), (9, 1000000)])
def test_phased_launch_equals_the_single_kernel(sfb, oracle, grid, slice_iters, knobs):
    """SFB_SP_PHASED=1 splits a time-sliced launch into three kernels (setup / ADMM loop / polish + report, each with
    its own grid; the item's state travels through its workspace header like a suspended item's).  Same bits as the
    single kernel: plain and pruned plans, cold and warm start, an item on the fallback path (solved completely in the
    setup phase and skipped by the later ones), items that fail the pre-check (iter == 0) and a max_iter cut-off."""
    variant, K, B = 6, 10, 150
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=23)
    keep = np.any(Av != 0.0, axis=0)
    Av[7, np.nonzero(~keep)[0][2]] = -0.25       # violates the mask: fallback pool
    l[11, 3], u[11, 3] = 1.0, -1.0                 # u < l: PrimalInfeasible at the pre-check, no iteration
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    for prm in (sfb.QPSolverParams(max_iter=4000), sfb.QPSolverParams(max_iter=60, polish=False)):
        for kp in (None, keep):
            plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=kp)
            knobs.set(SFB_SP_SLICE=str(slice_iters))
            knobs.set(SFB_SP_GRID=str(grid))
            knobs.set(SFB_SP_PHASED="0")
            base = plan.solve_batch_host(Px, q, Av, l, u, prm)
            base2 = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.5 * base.primal, warm_y=0.5 * base.dual)
            knobs.set(SFB_SP_PHASED="1")
            r = plan.solve_batch_host(Px, q, Av, l, u, prm)
            r2 = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.5 * base.primal, warm_y=0.5 * base.dual)
            for a, b in ((r, base), (r2, base2)):
                assert np.array_equal(a.code, b.code) and np.array_equal(a.iter, b.iter)
                assert np.array_equal(a.primal, b.primal, equal_nan=True) and np.array_equal(a.dual, b.dual, equal_nan=True)
                assert np.array_equal(a.objective, b.objective, equal_nan=True)
            assert r.code[11] == 2 and r.iter[11] == 0
            if prm.max_iter == 60:
                assert (r.code == 4).any() and r.iter.max() == 60
    knobs.clear("SFB_SP_SLICE", "SFB_SP_GRID", "SFB_SP_PHASED")


@pytest.mark.parametrize("grid,pause,lat", [(16, 27, 1), (48, 2, 0), (9, 60, 1), (5, 27, 0)])
def test_launch_in_predicted_order_equals_the_single_kernel(sfb, oracle, grid, pause, lat, knobs):
    """Default for time-sliced launches: a first launch takes every item through setup and its first `pause` iterations
    (items done by then are polished and reported there), the survivors leave a score -- residual over tolerance at
    their last stopping check -- and wait in their workspace; a counting sort orders them by descending score and a second
    launch finishes them longest-first.  A schedule only: same bits as the single kernel (SFB_SP_PREDICT=0) and as the
    oracle, for plain and pruned plans, cold and warm start, an item on the fallback path, a pre-check failure, a
    max_iter cut-off and another stop_check_iter; with the loop launch in the LAT form (loop vectors in LDS, chained sweeps:
    the default) and in the standard form (SFB_SP_LAT=0)."""
    variant, K, B = 6, 10, 150
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=29)
    keep = np.any(Av != 0.0, axis=0)
    Av[9, np.nonzero(~keep)[0][1]] = 0.5          # violates the mask: fallback pool, solved completely in the first launch
    l[13, 2], u[13, 2] = 1.0, -1.0                 # u < l: PrimalInfeasible at the pre-check, no iteration
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    knob_names = ("SFB_SP_GRID", "SFB_SP_PREDICT", "SFB_SP_PAUSE", "SFB_SP_LAT")
    for prm in (sfb.QPSolverParams(max_iter=4000), sfb.QPSolverParams(max_iter=150, polish=False),
                sfb.QPSolverParams(max_iter=4000, stop_check_iter=7)):
        for kp in (None, keep):
            plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=kp)
            knobs.set(SFB_SP_GRID=str(grid))
            knobs.set(SFB_SP_PREDICT="0")
            base = plan.solve_batch_host(Px, q, Av, l, u, prm)
            base2 = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.5 * base.primal, warm_y=0.5 * base.dual)
            knobs.clear("SFB_SP_PREDICT")
            knobs.set(SFB_SP_PAUSE=str(pause))
            knobs.set(SFB_SP_LAT=str(lat))
            r = plan.solve_batch_host(Px, q, Av, l, u, prm)
            r2 = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.5 * base.primal, warm_y=0.5 * base.dual)
            for a, b in ((r, base), (r2, base2)):
                assert np.array_equal(a.code, b.code) and np.array_equal(a.iter, b.iter)
                assert np.array_equal(a.primal, b.primal, equal_nan=True) and np.array_equal(a.dual, b.dual, equal_nan=True)
                assert np.array_equal(a.objective, b.objective, equal_nan=True)
            assert r.code[13] == 2 and r.iter[13] == 0
            assert (r.iter > pause + 25).any()                                   # the second launch had work ...
            assert pause < 27 or (r.iter[r.iter > 0] <= pause + 25).any()        # ... and some items were done in the first
            if prm.max_iter == 150:
                assert (r.code == 4).any() and r.iter.max() == 150
            if kp is None:
                ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(),
                                                   params=_oracle_params(oracle, prm), nthreads=8)
                assert np.array_equal(r.iter, ref["iter"]) and np.array_equal(r.code, ref["code"]) and np.array_equal(r.primal, ref["x"])
            knobs.clear(*knob_names[1:])
    knobs.clear("SFB_SP_GRID")


def test_max_time_on_the_sparse_path(sfb, oracle):
    """max_time (qp_solver.hpp:504-507) on the sparse kernel: 1 ns ends every agent at its first stopping check;
    deterministic and equal to the oracle with the same limit.  Also through a time-sliced launch."""
    variant, K, B = 6, 10, 40
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=4)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    prm = sfb.QPSolverParams(max_time=1e-9)
    r = plan.solve_batch_host(Px, q, Av, l, u, prm)
    op = _oracle_params(oracle, prm)
    op.max_time_ns = 1
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(), params=op, nthreads=8)
    assert np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"]) and np.array_equal(r.primal, ref["x"])
    assert (r.code == 5).all() and (r.iter == 2).all()


def test_mpc_closed_loop_like_reference_test(sfb):
    """tests/test_mpc.cpp:83-117: Optimal on consecutive calls, u1 ~ u2 ~ u3 with and without warm start,
    trajectory sizes."""
    u = np.zeros((6, 2)); codes = np.zeros(6, np.int32); sizes = np.zeros(4, np.int32)
    rc = M.lib().sfbx_test_mpc_se2(u.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                   sizes.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert (codes == 0).all()
    for blk in (u[:3], u[3:]):
        assert np.abs(blk[0] - blk[1]).max() < 1e-6 and np.abs(blk[0] - blk[2]).max() < 1e-6
    assert np.abs(u[0] - u[3]).max() < 1e-6
    assert sizes[0] + 1 == sizes[1] and sizes[2] + 1 == sizes[3]
    assert np.all(np.abs(u) <= 2.0 + 1e-6)  # udes (1) (+) du with |u_total| <= ... cr bounds total u in [-1, 1]


def test_reference_shaped_mpc_caller_code(sfb):
    """tests/test_mpc.cpp:60-155 transcribed against <smooth/feedback/mpc.hpp> with only the Lie types renamed
    (examples/models.cpp::sfbx_test_mpc_api): MPC<T, X, U, F, CR>, `mpc(1, x)`, `mpc(4, x, us, xs)`, functors held by
    reference that see set_time, copies / moves; plus the declaration of examples/mpc_asif_vehicle.cpp:64 on a
    std::chrono::duration clock.  Copies own their solvers and share the desired trajectories (mpc.hpp:407, 607-608)."""
    out = np.full(14, np.nan); codes = np.full(12, -1, np.int32)
    rc = M.lib().sfbx_test_mpc_api(out.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert (codes == 0).all(), codes                       # ASSERT_EQ(code, Optimal) everywhere
    assert out[0] < 1e-6 and out[1] < 1e-6                 # u1 ~ u2 (warm start), u3 ~ u1
    assert out[2] == 1.0                                   # us.size() + 1 == xs.size()
    assert out[3] >= 4.0 and out[4] >= 4.0                 # ASSERT_GE(f.t_, 4), ASSERT_GE(cr.t_, 4)
    assert out[5] < 1e-6                                   # pointer overload (warm-started from the solve before)
    assert (out[6:10] < 1e-6).all(), out[6:10]             # copy / copy-assign / move / move-assign
    assert out[10] <= 0.5 + 1e-9                           # the vehicle's input inside its box
    assert out[11] == 1.0                                  # every copy analysed for itself
    assert out[12] < 1e-6 and out[13] > 1e-3               # a setter on a copy reaches the original


def test_mpc_swarm_tick(sfb):
    """MPCSwarm: host assembly on threads + ONE batched GPU solve per tick, warm-started ticks."""
    B = 64
    u0 = np.zeros((B, 2)); codes = np.zeros(B, np.int32); iters = np.zeros(B, np.uint32)
    rc = M.lib().sfbx_mpc_swarm_step(6, 30, C.c_double(5.0), C.c_int64(B), C.c_uint64(1), 3,
                                     u0.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                     iters.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert (codes == 0).all()
    assert np.all(np.abs(u0) <= 0.5 + 1e-6)


def test_launch_order_does_not_change_results(sfb):
    """sfb_sparse_qp_solve_batch_ordered: any permutation of the launch order gives the same solutions, bit for bit
    (the MPC swarm launches last tick's long-running agents first)."""
    import torch
    variant, K, B = 6, 30, 200
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=5)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    prm = sfb.QPSolverParams(max_iter=4000)
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
    ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
    out = []
    rng = np.random.default_rng(0)
    for order in (None, np.arange(B, dtype=np.int32)[::-1].copy(), rng.permutation(B).astype(np.int32)):
        x = torch.full((B, d["n"]), np.nan, dtype=torch.float64, device=dev)
        y = torch.full((B, d["m"]), np.nan, dtype=torch.float64, device=dev)
        it = torch.zeros(B, dtype=torch.int32, device=dev); code = torch.full((B,), -1, dtype=torch.int32, device=dev)
        dord = T(order) if order is not None else None
        plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(),
                                y.data_ptr(), 0, it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm,
                                stream=torch.cuda.current_stream().cuda_stream, dorder=dord.data_ptr() if dord is not None else 0)
        torch.cuda.synchronize()
        out.append((x.cpu().numpy(), y.cpu().numpy(), it.cpu().numpy(), code.cpu().numpy()))
    for o in out[1:]:
        for a, b in zip(out[0], o):
            assert np.array_equal(a, b)
    assert (out[0][3] == 0).all()


class _DeviceSolver:
    """sfb_sparse_qp_solve_batch on a workspace that lives across calls (what the MPC swarm does every tick)."""

    def __init__(self, sfb, plan, B, poison=None):
        import torch
        self.sfb, self.plan, self.B, self.torch = sfb, plan, B, torch
        self.dev = torch.device("cuda:0")
        self.ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=self.dev)
        if poison is not None:
            self.ws.fill_(poison)

    def __call__(self, Px, q, Ax, l, u, prm, warm=None):
        torch, B, plan = self.torch, self.B, self.plan
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.dev)
        d = [T(a) for a in (Px, q, Ax, l, u)]
        w = [T(a) for a in warm] if warm is not None else None
        x = torch.full((B, plan.n), np.nan, dtype=torch.float64, device=self.dev)
        y = torch.full((B, plan.m), np.nan, dtype=torch.float64, device=self.dev)
        obj = torch.full((B,), np.nan, dtype=torch.float64, device=self.dev)
        it = torch.zeros(B, dtype=torch.int32, device=self.dev)
        code = torch.full((B,), -1, dtype=torch.int32, device=self.dev)
        plan.solve_batch_device(B, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(),
                                code.data_ptr(), self.ws.data_ptr(), prm, dwarm_x=w[0].data_ptr() if w else 0,
                                dwarm_y=w[1].data_ptr() if w else 0, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return tuple(a.cpu().numpy() for a in (code, it, x, y, obj))


@pytest.mark.parametrize("pruned", [False, True])
@pytest.mark.parametrize("polish", [True, False])
def test_factor_reuse_gives_the_same_bits(sfb, pruned, polish, knobs):
    """sfb_qp_params::reuse_factor (f3, the exact case): the caller vouches that P and A are unchanged since the
    previous call on the workspace; the kernel keeps the compacted A, the scaling and the LDL' factor wherever they
    are provably what it would recompute (c and the rho vector are re-derived and compared) and recomputes otherwise.
    A sequence of ticks on ONE workspace with the flag set is compared, bit for bit, with the same problems solved
    without the flag on a fresh workspace: q, l, u moving (everything kept), |q| large enough to change c (scaling
    redone), an inequality row turning into an equality (rho changes: factor redone on the kept scaling), a tick
    with a new A and no flag, and a flagged tick after it.  With polish the ADMM factor has to survive the polish
    factorisation (second factor block of the workspace).  Small grid: the launches are time-sliced, items are
    suspended and resumed by other blocks in between."""
    knobs.set(SFB_SP_GRID="16")
    variant, K, B = 6, 20, 48
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=21)
    Av2, _, _ = M.mpc_assemble_batch(variant, K, B, seed=22)
    keep = np.any(np.concatenate([Av, Av2]) != 0.0, axis=0) if pruned else None
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    rng = np.random.default_rng(3)
    Px = np.tile(Pv, (B, 1))
    n, m = d["n"], d["m"]
    ineq = l < u
    q_small = 1e-3 * rng.uniform(-1, 1, (B, n))
    q_big = q_small * 1e4                                        # |q| beyond the mean column norm of P: c changes
    l_eq, u_eq = l.copy(), u.copy()
    row = np.nonzero(ineq[0])[0][:3]
    l_eq[::2, row] = u_eq[::2, row] = 0.1                        # inequality rows of every other agent become equalities
    ticks = [                                                    # (A, q, l, u, flag)
        (Av, q_small, l, u, True),                               # first call: nothing to keep (workspace is garbage)
        (Av, -q_small, l - 0.01 * ineq, u + 0.02 * ineq, True),  # everything kept
        (Av, q_big, l, u, True),                                 # c changes
        (Av, q_big, l_eq, u_eq, True),                           # rho changes for half of the agents
        (Av2, q_small, l, u, False),                             # new A: the caller does not set the flag
        (Av2, q_small * 0.5, l, u + 0.01 * ineq, True),          # and may set it again afterwards
    ]
    run = _DeviceSolver(sfb, plan, B, poison=float("nan"))
    warm = None
    for t, (A_, q_, l_, u_, flag) in enumerate(ticks):
        prm = sfb.QPSolverParams(max_iter=4000, polish=polish, reuse_factor=flag)
        got = run(Px, q_, A_, l_, u_, prm, warm)
        fresh = _DeviceSolver(sfb, plan, B, poison=0.0)
        ref = fresh(Px, q_, A_, l_, u_, sfb.QPSolverParams(max_iter=4000, polish=polish), warm)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True), "tick %d" % t
        assert (got[0] == 0).mean() > 0.9
        warm = (got[2], got[3])


def test_factor_reuse_really_skips_the_work(sfb):
    """White box: with a pruned plan the kernel works on its own compacted copy of A.  A flagged call that is handed
    GARBAGE in the kept entries of A (against the contract: the masked entries stay zero for the guard) must still
    return the true problem's solution -- everything that depends on A (compaction, scaling, factor, residuals) came
    from the workspace.  The same call without the flag solves the garbage problem instead."""
    variant, K, B = 6, 20, 16
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=21)
    keep = np.any(Av != 0.0, axis=0)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    run = _DeviceSolver(sfb, plan, B)
    prm = sfb.QPSolverParams(max_iter=4000, reuse_factor=True)
    first = run(Px, q, Av, l, u, prm)
    garbage = Av * keep * 1.5 + 0.25 * keep
    again = run(Px, q, garbage, l, u, prm)
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    other = run(Px, q, garbage, l, u, sfb.QPSolverParams(max_iter=4000))
    assert not np.array_equal(other[2], first[2])


def test_double_integrator_mpc_reuses_its_factor(sfb):
    """examples/mpc_doubleintegrator.cpp (x = (p, v), |u| <= 0.5, K = 20, tf = 5, 50 ms ticks) in closed loop: a linear
    system, so the QP matrices of consecutive ticks are identical; the C++ solver front (QPSolver::solve) recognises
    that and flags reuse_factor from the second tick on.  Inputs and iteration counts equal those of the same loop
    with the reuse switched off, bit for bit; the loop converges to the desired trajectory within the input limit."""
    r = M.mpc_doubleintegrator(40)
    assert r["reuse_count"] == 39
    assert np.array_equal(r["u"], r["u_ref"]) and np.array_equal(r["iter"], r["iter_ref"])
    assert (r["code"] == 0).all()
    assert np.abs(r["u"]).max() <= 0.5 + 1e-3 and np.abs(r["u"][0]) > 0.4    # starts 0.6 off: saturated input
    print("double integrator closed loop, 40 ticks: %.1f ms with factor reuse, %.1f ms without" % tuple(1e3 * r["seconds"]))


def test_generic_ocp_to_qp_solves_and_maps_back(sfb):
    """examples/ocp_se2_qp.cpp:33-48 flow on the problem of tests/test_ocp_to_qp.cpp: ocp_to_qp() -> solve_qp (sparse
    kernel) -> qpsol_to_ocpsol().  The QP is in deviations from (xl, ul) and nothing pins x_0, so the end-point cost
    decides dx_N: the reference enters HALF the Hessian of theta into P (ocp_to_qp.hpp:191-193, block_add(..., 0.5))
    next to the full gradient in q (:195-196), i.e. it minimises 1/2 dx' (1/2 H) dx + g' dx with H = 2 I,
    g = 2 xl(tf): dx_N = -2 xl(tf) and x(tf) = -xl(tf) = (-0.2, -0.2) -- reproduced as is.  Optimal, input within
    |u| <= 1; the returned trajectories are the linearisation plus the interpolated deviations (exact at nodes)."""
    o = M.ocp_to_qp_basic(solve=True)
    assert o[12] == 0                        # Optimal
    assert abs(o[13]) <= 1.0 + 1e-3          # input within the running constraint
    assert abs(o[14] + 0.2) < 1e-2 and abs(o[15] + 0.2) < 1e-2
    assert abs(o[17]) < 1e-12


def test_factor_reuse_after_an_ordered_launch_is_safe(sfb):
    """A launch with an explicit order uses workspace slot = launch position, not item: it must not leave anything a
    later natural-order call with reuse_factor could mistake for the items' own factors."""
    import torch
    variant, K, B = 6, 20, 40
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=31)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    run = _DeviceSolver(sfb, plan, B)
    prm = sfb.QPSolverParams(max_iter=4000, reuse_factor=True)
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    order = T(np.arange(B, dtype=np.int32)[::-1].copy())
    dd = [T(a) for a in (Px, q, Av, l, u)]
    x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev); code = torch.zeros(B, dtype=torch.int32, device=dev)
    plan.solve_batch_device(B, *[a.data_ptr() for a in dd], x.data_ptr(), y.data_ptr(), 0, it.data_ptr(), code.data_ptr(),
                            run.ws.data_ptr(), prm, stream=torch.cuda.current_stream().cuda_stream, dorder=order.data_ptr())
    torch.cuda.synchronize()
    got = run(Px, q, Av, l, u, prm)                       # natural order, flagged: slot b now holds item B-1-b's leftovers
    ref = _DeviceSolver(sfb, plan, B)(Px, q, Av, l, u, sfb.QPSolverParams(max_iter=4000))
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    assert np.array_equal(x.cpu().numpy(), ref[2])


def test_lat_loop_launch_with_resident_stream_and_polishers_at_the_headline_model(sfb, oracle, knobs):
    """The headline model's plan (nx = 12, nu = 2, K = 50: forward stream of 168 units) through the launch in predicted order on
    a tiny grid: the LAT loop launch keeps the first 64 units of the forward factor stream in its AccVGPRs (hand-scheduled
    prefix with literal registers) and the polishers take finished items off its ring -- against the same launch without
    polishers, with the standard-form loop launch, as a single time-sliced launch, and against the oracle: same bits."""
    variant, K, B = 12, 50, 72
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=43)
    keep = np.any(Av != 0.0, axis=0)
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    prm = sfb.QPSolverParams()
    knobs.set(SFB_SP_GRID=12)
    res = {}
    for name, kn in (("default", {}), ("no_polishers", {"SFB_SP_POLISHERS": 0}), ("two_polishers", {"SFB_SP_POLISHERS": 2}),
                     ("standard_loop", {"SFB_SP_LAT": 0}), ("single_launch", {"SFB_SP_PREDICT": 0}), ("pause_27", {"SFB_SP_PAUSE": 27})):
        knobs.set(**kn)
        res[name] = plan.solve_batch_host(Px, q, Av, l, u, prm)
        res[name + "_warm"] = plan.solve_batch_host(Px, q, Av, l, u, prm, warm_x=0.3 * res[name].primal, warm_y=0.3 * res[name].dual)
        knobs.clear(*kn)
    base = res["single_launch"]
    for name, r in res.items():
        b = res["single_launch_warm"] if name.endswith("_warm") else base
        assert np.array_equal(r.code, b.code) and np.array_equal(r.iter, b.iter), name
        assert np.array_equal(r.primal, b.primal) and np.array_equal(r.dual, b.dual) and np.array_equal(r.objective, b.objective), name
    assert base.iter.max() > 100 and (base.code == 0).all()
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(),
                                       params=_oracle_params(oracle, prm), nthreads=8)
    assert np.array_equal(base.iter, ref["iter"]) and np.array_equal(base.code, ref["code"]) and np.array_equal(base.primal, ref["x"])


def test_two_plans_on_two_streams_concurrently(sfb, oracle, knobs):
    """Two DIFFERENT plans (the headline model and the README's vehicle), each on its own HIP stream with its own buffers,
    several solves enqueued back to back without a synchronisation in between: every caller stream has its own polishers'
    stream, hand-off events and bit in the device's busy word (csrc/qp_sparse.hip SparseDeviceBook), the helpers of one launch
    leave when the other stream has work.  Every result of every round equals the oracle's, bit for bit."""
    import torch
    dev = torch.device("cuda:0")
    knobs.set(SFB_SP_GRID=12)  # both batches go through the launch in predicted order (LAT loop launch + polishers)
    prm = sfb.QPSolverParams()
    jobs = []
    for variant, K, B, seed in ((12, 50, 72, 47), (6, 30, 60, 48)):
        d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
        Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=seed)
        keep = np.any(Av != 0.0, axis=0)
        Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
        plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
        ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm, forder=plan.factor_order(),
                                           params=_oracle_params(oracle, prm), nthreads=8)
        f64 = dict(dtype=torch.float64, device=dev)
        din = [torch.tensor(a, **f64) for a in (Px, q, Av, l, u)]
        rounds = []
        for _ in range(3):  # own outputs and workspace per round: nothing is overwritten before it is compared
            rounds.append(dict(x=torch.empty((B, d["n"]), **f64), y=torch.empty((B, d["m"]), **f64), obj=torch.empty(B, **f64),
                               it=torch.empty(B, dtype=torch.int32, device=dev), code=torch.empty(B, dtype=torch.int32, device=dev),
                               ws=torch.empty(plan.workspace_bytes(B), dtype=torch.uint8, device=dev)))
        jobs.append(dict(plan=plan, B=B, din=din, rounds=rounds, ref=ref, stream=torch.cuda.Stream()))
    torch.cuda.synchronize()
    for r in range(3):
        for j in jobs:  # alternating: both streams always have a solve enqueued
            o = j["rounds"][r]
            j["plan"].solve_batch_device(j["B"], *[t.data_ptr() for t in j["din"]], o["x"].data_ptr(), o["y"].data_ptr(),
                                         o["obj"].data_ptr(), o["it"].data_ptr(), o["code"].data_ptr(), o["ws"].data_ptr(), prm,
                                         stream=j["stream"].cuda_stream)
    torch.cuda.synchronize()
    for j in jobs:
        ref = j["ref"]
        assert ref["iter"].max() > 50
        for o in j["rounds"]:
            assert np.array_equal(o["code"].cpu().numpy(), ref["code"])
            assert np.array_equal(o["it"].cpu().numpy().astype(np.uint32), ref["iter"])
            assert np.array_equal(o["x"].cpu().numpy(), ref["x"]) and np.array_equal(o["y"].cpu().numpy(), ref["y"])
            assert np.array_equal(o["obj"].cpu().numpy(), ref["obj"])
