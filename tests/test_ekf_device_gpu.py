"""A swarm of EKFs resident on the GPU (include/smooth_feedback_amd/ekf_device.hpp): linearisation, state step and
g (+) delta run in device code next to the batched covariance kernels.  Checked against one host EKF<> object per
filter (include/smooth_feedback_amd/ekf.hpp: the reference's predict / update semantics, ekf.hpp:79-103, :116-139).
Needs an MI355X."""
import numpy as np
import pytest

from examples import models_lib as M

pytestmark = pytest.mark.gpu

# the two fronts run the same helper functions and the same covariance kernels; what differs is sin / cos / atan2 of
# the host and device maths libraries in the group operations, amplified by the forward-difference step 1.5e-8
TOL = 1e-6


@pytest.mark.parametrize("rk4,dt", [(False, 0.0), (False, 0.03), (True, 0.0), (True, 0.04)])
def test_swarm_equals_one_host_filter_per_agent(rk4, dt):
    st, P0, y = M.ekf_swarm_inputs(24, 3, seed=5)
    dev = M.ekf_swarm_device(st, P0, y, tau=0.1, dt=dt, rk4=rk4)
    host = M.ekf_swarm_host(st, P0, y, tau=0.1, dt=dt, rk4=rk4)
    assert np.all(dev["info"] == 0)
    assert np.all(np.isfinite(dev["states"])) and np.all(np.isfinite(dev["P"]))
    assert np.max(np.abs(dev["states"] - host["states"])) <= TOL
    assert np.max(np.abs(dev["P"] - host["P"])) <= TOL
    assert np.max(np.abs(dev["states"] - st)) > 1e-3  # the filters moved


def test_fused_step_equals_predict_then_update():
    """step() = predict with one substep + update, the covariance passes in one launch: same bits"""
    st, P0, y = M.ekf_swarm_inputs(1000, 4, seed=2)
    a = M.ekf_swarm_device(st, P0, y, tau=0.05, fused=True)
    b = M.ekf_swarm_device(st, P0, y, tau=0.05, fused=False)
    assert np.array_equal(a["states"], b["states"]) and np.array_equal(a["P"], b["P"])


def test_one_launch_round_equals_the_separate_launches():
    """step() runs linearisation, covariance step, state step, Kalman update and g (+) delta of a round in ONE kernel
    (P crosses HBM once each way; A, H, r, delta stay in registers); the same round as four launches on resident
    buffers, or with the measurements already on the device, gives the same bits."""
    st, P0, y = M.ekf_swarm_inputs(3000, 5, seed=4)
    one = M.ekf_swarm_device(st, P0, y, tau=0.05, fused=1)
    four = M.ekf_swarm_device(st, P0, y, tau=0.05, fused=2)
    res = M.ekf_swarm_device(st, P0, y, tau=0.05, fused=3)
    plain = M.ekf_swarm_device(st, P0, y, tau=0.05, fused=0)
    for other in (four, res, plain):
        assert np.array_equal(one["states"], other["states"]) and np.array_equal(one["P"], other["P"])
        assert np.array_equal(one["info"], other["info"])
    assert np.max(np.abs(one["states"] - st)) > 1e-3


def test_covariances_stay_symmetric_and_positive():
    st, P0, y = M.ekf_swarm_inputs(512, 6, seed=9)
    r = M.ekf_swarm_device(st, P0, y, tau=0.1, dt=0.025)
    P = r["P"].reshape(-1, 6, 6)
    assert np.array_equal(P, P.transpose(0, 2, 1))  # symU mirrors (ekf.hpp:88, :138)
    assert np.min(np.linalg.eigvalsh(P)) > 0
    # measured components are known better than before
    assert np.mean(P[:, 0, 0]) < np.mean(P0.reshape(-1, 6, 6)[:, 0, 0])


@pytest.mark.parametrize("batch", [1, 65])
def test_ragged_swarm_sizes(batch):
    st, P0, y = M.ekf_swarm_inputs(batch, 2, seed=11)
    dev = M.ekf_swarm_device(st, P0, y, tau=0.1, dt=0.05)
    host = M.ekf_swarm_host(st, P0, y, tau=0.1, dt=0.05)
    assert np.max(np.abs(dev["states"] - host["states"])) <= TOL and np.max(np.abs(dev["P"] - host["P"])) <= TOL
