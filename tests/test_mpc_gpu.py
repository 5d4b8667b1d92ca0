"""MPC path on the GPU: QPs produced by the host-side MPC transcription (C++ front, examples/models.cpp)
solved by the sparse HIP kernel, checked against the sparse CPU oracle; closed-loop behaviour of
tests/test_mpc.cpp.  Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest

import models_lib as M
from test_qp_dense_gpu import _compare, _oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "masked", "plain"])
def sweep_mode(request):
    """The sparse kernel streams the factor with plain loads or, while many waves are resident, with loads masked
    to the lanes that carry entries (no HBM traffic for the padding of the sweep schedule).  'auto' is the
    product's choice (plain for these small batches); the other two force one form (SFB_SP_LEAN_WAVES is read at
    every launch).  All three must agree with the oracle bit for bit."""
    import os
    old = os.environ.get("SFB_SP_LEAN_WAVES")
    if request.param != "auto":
        os.environ["SFB_SP_LEAN_WAVES"] = "-1" if request.param == "masked" else "1000000000"
    yield request.param
    if old is None:
        os.environ.pop("SFB_SP_LEAN_WAVES", None)
    else:
        os.environ["SFB_SP_LEAN_WAVES"] = old


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
    ref = oracle.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=plan.perm,
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
                                        perm=plan.perm, params=_oracle_params(oracle, prm), warm_x=ref["x"],
                                        warm_y=ref["y"], nthreads=8)
    _compare(r2, ref2)
    assert r2.iter.mean() <= r.iter.mean()


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
    ws = torch.empty(B * plan.workspace_bytes_per_item // 8, dtype=torch.float64, device=dev)
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
