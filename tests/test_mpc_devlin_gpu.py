"""MPC swarm with the linearisation on the GPU (include/smooth_feedback_amd/mpc_device.hpp -> sfb_mpc_swarm_device_records /
sfb_mpc_swarm_step_resident): the records the device kernel writes against MPC::fill_record on the host, and the closed
loop against MPCSwarmDevice (host linearisation, same kernels downstream).  Needs an MI355X."""
import numpy as np
import pytest

from examples import models_lib as M

pytestmark = pytest.mark.gpu

# the records differ by what sin / cos of the host and device maths libraries differ (a few ulp); everything after the
# records is the same device code
REC_TOL = 1e-12
U_TOL = 1e-6  # north_star's fp64 tolerance is 1e-8 on a QP's solution; u0 here is 3 closed-loop ticks downstream


@pytest.mark.parametrize("variant,K", [(6, 20), (12, 20), (12, 50)])
def test_device_records_equal_fill_record(variant, K):
    """one tick; unpacked (the fallback layout) and packed records, entry by entry"""
    batch = 96
    L, host = M.mpc_records(variant, K, batch, seed=1)
    full = M.mpc_swarm_devlin_step(variant, K, batch, 1, seed=1, probe_empty=True)
    assert not full["packed"] and full["record_doubles"] == L.record_doubles()
    assert np.max(np.abs(full["records"] - host)) <= REC_TOL
    pk = M.mpc_swarm_devlin_step(variant, K, batch, 1, seed=1)
    assert pk["packed"] and pk["record_doubles"] < L.record_doubles()
    Lp = M.mpc_layout(variant, K)
    Lp.jac_keep = Lp.jac_keep_of(host)
    want = Lp.pack_records(host)
    assert want.shape == pk["records"].shape
    assert np.max(np.abs(pk["records"] - want)) <= REC_TOL
    # the two record layouts describe the same QPs: identical solves
    assert np.array_equal(pk["code"], full["code"]) and np.array_equal(pk["iter"], full["iter"])
    assert np.array_equal(pk["u0"], full["u0"])


@pytest.mark.parametrize("variant,K,batch", [(6, 20, 200), (12, 20, 500), (12, 50, 64), (12, 20, 1), (6, 20, 65)])
def test_closed_loop_equals_host_linearised_swarm(variant, K, batch):
    ticks = 3
    u_ref, c_ref, it_ref = M.mpc_swarm_step(variant, K, batch, ticks, seed=1, device=True)
    r = M.mpc_swarm_devlin_step(variant, K, batch, ticks, seed=1, want_records=False)
    assert np.array_equal(r["code"], c_ref)
    assert np.all(r["code"] == 0)
    assert np.max(np.abs(r["u0"] - u_ref)) <= U_TOL
    # iteration counts may move by a check interval where a residual sits on the tolerance
    assert np.mean(r["iter"] == it_ref) >= (0.95 if batch >= 64 else 0.0)


def test_misfit_falls_back_to_unpacked_records():
    """a probe that saw nothing: the first tick flags the misfit on the device, the swarm switches layout and repeats"""
    a = M.mpc_swarm_devlin_step(12, 20, 128, 2, seed=3, probe_empty=True, want_records=False)
    b = M.mpc_swarm_devlin_step(12, 20, 128, 2, seed=3, want_records=False)
    assert not a["packed"] and b["packed"]
    assert np.array_equal(a["code"], b["code"]) and np.array_equal(a["iter"], b["iter"]) and np.array_equal(a["u0"], b["u0"])
