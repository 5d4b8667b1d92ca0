"""Several devices from ONE process (include/sfb.h: sfb_set_devices, sfb_*_solve_batch_host_multi; C++ front:
QPSolver::shard_over_devices): the batch is cut into contiguous shards, one host thread / plan upload / workspace per
device, host memory is the gathering point.  The GPU boxes of the test tier have one device, so the device list names
ordinal 0 several times -- every piece of the code path runs (sharding, per-shard pointers, threads, per-device state,
error propagation) except two physical devices working at the same time."""
import numpy as np
import pytest

from examples import models_lib as M

pytestmark = pytest.mark.gpu


@pytest.fixture()
def three_shards(sfb):
    sfb._capi.set_devices([0, 0, 0])
    assert sfb._capi.get_devices() == [0, 0, 0]
    yield
    sfb._capi.set_devices(None)
    assert sfb._capi.get_devices() == list(range(sfb._capi.device_count()))


@pytest.mark.parametrize("B", [1, 2, 100, 1000])
def test_dense_batch_sharded_equals_single_device(sfb, three_shards, B):
    """Dense QPs (n = 10, m = 20): batches smaller than the device list (empty shards), not divisible by it, cold and
    warm start -- bit-identical to the single-device call."""
    P, q, A, l, u = sfb.random_qp_batch(9, B, 20, 10, 1.0)
    prm = sfb.QPSolverParams(max_iter=2000)
    one = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    many = sfb.solve_qp_batch_host(P, q, A, l, u, prm, multi_device=True)
    wone = sfb.solve_qp_batch_host(P, q, A, l, u, prm, warm_x=0.5 * one.primal, warm_y=0.5 * one.dual)
    wmany = sfb.solve_qp_batch_host(P, q, A, l, u, prm, warm_x=0.5 * one.primal, warm_y=0.5 * one.dual, multi_device=True)
    for a, b in ((one, many), (wone, wmany)):
        assert np.array_equal(a.code, b.code) and np.array_equal(a.iter, b.iter)
        assert np.array_equal(a.primal, b.primal, equal_nan=True) and np.array_equal(a.dual, b.dual, equal_nan=True)
        assert np.array_equal(a.objective, b.objective, equal_nan=True)


@pytest.mark.parametrize("B", [2, 77])
def test_sparse_batch_sharded_equals_single_device(sfb, three_shards, B):
    """MPC QPs through a pruned plan (with an item on the fallback path), reuse_factor requested: same bits."""
    variant, K = 6, 10
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=31)
    keep = np.any(Av != 0.0, axis=0)
    Av[B - 1, np.nonzero(~keep)[0][1]] = 0.5
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    for prm in (sfb.QPSolverParams(max_iter=4000), sfb.QPSolverParams(max_iter=4000, reuse_factor=True)):
        one = plan.solve_batch_host(Px, q, Av, l, u, prm)
        many = plan.solve_batch_host(Px, q, Av, l, u, prm, multi_device=True)
        again = plan.solve_batch_host(Px, q, Av, l, u, prm, multi_device=True)
        for b in (many, again):
            assert np.array_equal(one.code, b.code) and np.array_equal(one.iter, b.iter)
            assert np.array_equal(one.primal, b.primal) and np.array_equal(one.dual, b.dual)


def test_error_of_a_shard_reaches_the_caller(sfb, three_shards):
    """A failure inside a shard (here: max_iter beyond uint32, rejected by the per-shard argument check) comes back
    as the call's status with the device and the message."""
    P, q, A, l, u = sfb.random_qp_batch(9, 6, 20, 10, 1.0)
    with pytest.raises(sfb._capi.SfbError) as ei:
        sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=2 ** 40), multi_device=True)
    assert "max_iter" in str(ei.value)
    with pytest.raises(sfb._capi.SfbError):
        sfb._capi.set_devices([0, 99])


def test_cpp_swarm_sharded_over_devices_equals_single_device(sfb):
    """MPCSwarm (C++ front, host assembly) with QPSolver::shard_over_devices over {0, 0, 0, 0}: two closed-loop ticks
    (cold start, then warm start) of 150 vehicles give the inputs, codes and iteration counts of the plain swarm."""
    one = M.mpc_swarm_step(6, 10, 150, 2, seed=5)
    many = M.mpc_swarm_step_multi(6, 10, 150, 2, [0, 0, 0, 0], seed=5)
    for a, b in zip(one, many):
        assert np.array_equal(a, b)
    assert (one[1] == 0).all() and one[2].max() > 25


@pytest.mark.parametrize("devices,threads", [([0, 0, 0], False), ([0, 0], True), ([0] * 7, True)])
def test_resident_mpc_swarm_sharded_over_devices_equals_single_device(sfb, devices, threads):
    """MPCSwarmMultiDeviceLin (multi_device.hpp): one resident swarm per shard (own device memory, plan upload, workspace),
    states up and u0 / code / iter down per tick.  Three closed-loop ticks (cold start, two warm starts) of 151 vehicles --
    not divisible by the shard counts -- equal the single-device MPCSwarmDeviceLin bit for bit; with a host thread per
    shard the shards use the device concurrently, as the threads of several devices would."""
    one = M.mpc_swarm_devlin_step(6, 10, 151, 3, seed=5, want_records=False)
    many = M.mpc_swarm_devlin_step_multi(6, 10, 151, 3, devices, seed=5, thread_per_shard=threads)
    for key in ("u0", "code", "iter"):
        assert np.array_equal(one[key], many[key]), key
    assert (one["code"] == 0).all() and one["iter"].max() >= 2


def test_resident_mpc_swarm_with_fewer_agents_than_devices(sfb):
    one = M.mpc_swarm_devlin_step(6, 10, 2, 2, seed=3, want_records=False)
    many = M.mpc_swarm_devlin_step_multi(6, 10, 2, 2, [0, 0, 0, 0, 0], seed=3)
    for key in ("u0", "code", "iter"):
        assert np.array_equal(one[key], many[key]), key


@pytest.mark.parametrize("rk4,fused,threads", [(False, False, False), (False, True, True), (True, False, True)])
def test_resident_ekf_swarm_sharded_over_devices_equals_single_device(sfb, rk4, fused, threads):
    """EKFSwarmMultiDevice: 1 003 vehicle filters over {0, 0, 0}, three predict + update rounds (Euler substeps / one-launch
    step() / RK4): estimates, covariances and the update info of the single-device EKFSwarmDevice, bit for bit."""
    st, P0, y = M.ekf_swarm_inputs(1003, 3, seed=2)
    kw = dict(tau=0.1, dt=0.0 if fused else 0.04, rk4=rk4, fused=fused)
    one = M.ekf_swarm_device(st, P0, y, **kw)
    many = M.ekf_swarm_device_multi(st, P0, y, [0, 0, 0], thread_per_shard=threads, **kw)
    for key in ("states", "P", "info"):
        assert np.array_equal(one[key], many[key]), key
