"""ASI filter front (include/smooth_feedback_amd/asif.hpp) on the GPU: the reference's own filter test, and the QPs
the front hands to libsfb.so re-solved by the dense CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest

from examples import models_lib as M

pytestmark = pytest.mark.gpu


def _oracle_solve(oracle, r, batch=False, polish=False, warm=None):
    prm = oracle.default_params(polish=int(polish))
    two = (lambda a: a) if batch else (lambda a: a[None, :])
    kw = {}
    if warm is not None:
        kw = dict(warm_x=two(warm[0]), warm_y=two(warm[1]))
    return oracle.qp_dense_solve_batch(two(r["P"]), two(r["q"]), two(r["A"]), two(r["l"]), two(r["ub"]), params=prm,
                                       nthreads=8, **kw)


def test_filter_so3_reference_case(oracle):
    """tests/test_asif.cpp:103-131: K = 100, nh = 3 (n = 4, m = 301) -> Optimal; the QP the front assembled, solved by
    the pivoted dense LDL' kernel for n + m > 64, equals the dense oracle bit for bit."""
    r = M.test_asif(0)
    assert (r["n"], r["m"]) == (4, 301)
    assert r["code"] == 0
    assert np.all(np.isfinite(r["u"]))
    ref = _oracle_solve(oracle, r, polish=True)
    assert r["code"] == ref["code"][0] and r["iter"] == ref["iter"][0]
    assert np.array_equal(r["x"], ref["x"][0]) and np.array_equal(r["y"], ref["y"][0])


def test_vehicle_default_size_is_bit_identical_to_the_dense_oracle(oracle):
    """K = 10: n = 3, m = 13 -> the dense kernel; same bits as the oracle on the QP the front assembled."""
    r = M.test_asif(2)
    assert (r["n"], r["m"]) == (3, 13)
    ref = _oracle_solve(oracle, r)
    assert r["code"] == ref["code"][0] == 0
    assert r["iter"] == ref["iter"][0]
    assert np.array_equal(r["x"], ref["x"][0]) and np.array_equal(r["y"], ref["y"][0])
    # the filter must brake: the nominal input drives towards the obstacle
    assert r["u"][0] < 0.4 - 1e-3 and -0.2 - 1e-6 <= r["u"][0] and abs(r["u"][1]) <= 0.5 + 1e-6


def test_vehicle_example_size_is_bit_identical_to_the_dense_oracle(oracle):
    """examples/mpc_asif_vehicle.cpp:105-129: K = 200 (n = 3, m = 203, polish off): the pivoted dense LDL' kernel for
    n + m > 64 -- same bits as the dense oracle on the QP the front assembled."""
    r = M.test_asif(1)
    assert (r["n"], r["m"]) == (3, 203)
    ref = _oracle_solve(oracle, r)
    assert r["code"] == ref["code"][0] == 0
    assert r["iter"] == ref["iter"][0]
    assert np.array_equal(r["x"], ref["x"][0]) and np.array_equal(r["y"], ref["y"][0])


@pytest.mark.parametrize("ticks", [1, 3])
def test_swarm_is_bit_identical_to_the_dense_oracle(oracle, ticks):
    """1 000 vehicles, K = 10 (k = 16, four QPs per wavefront), cold and warm-started ticks."""
    B = 1000
    r = M.asif_swarm_step(B, 10, ticks=ticks, seed=5)
    ref = _oracle_solve(oracle, r, batch=True, warm=(r["wx"], r["wy"]))
    assert np.array_equal(r["code"], ref["code"])
    assert np.array_equal(r["iter"], ref["iter"])
    assert np.array_equal(r["x"], ref["x"]) and np.array_equal(r["y"], ref["y"])
    assert (r["code"] == 0).mean() > 0.95
    ok = r["code"] == 0                   # input box of the example, to the ADMM tolerance (polish is off)
    assert np.all(r["u"][ok, 0] <= 0.5 + 1e-2) and np.all(r["u"][ok, 0] >= -0.2 - 1e-2)
    assert np.all(np.abs(r["u"][ok, 1]) <= 0.5 + 1e-2)
    if ticks > 1:
        assert np.abs(r["wx"]).max() > 0      # the last tick really was warm-started


@pytest.mark.parametrize("K,B", [(10, 300), (40, 64), (200, 16)])
def test_device_side_assembly_of_the_swarm(oracle, K, B):
    """ASIFSwarmDevice (asif_device.hpp): one GPU thread integrates one agent's backup trajectory and sensitivity
    (asif_func.hpp:145-179, the same function the host front runs) and writes its QP; the batched solve follows on the
    device.  (a) The QPs equal the host assembly's up to what the two maths libraries' sin / cos / atan2 differ
    (1e-9 relative here); (b) the solve of the device-assembled QPs is bit-identical to the dense oracle on those QPs
    (all dense kernels: k = 16, 46, 206); (c) in closed loop -- three ticks, warm-started -- the filtered inputs agree
    with the host front's."""
    st, ud = M.asif_swarm_states(B, seed=3)
    dev = M.asif_swarm_device_step(st, ud, K, ticks=1)
    host = M.asif_swarm_step(B, K, ticks=1, seed=3)
    for key in ("P", "q", "A", "l", "ub"):
        a, b = dev[key], host[key]
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin) and np.array_equal(a[~fin], b[~fin]), key
        assert np.abs(a[fin] - b[fin]).max(initial=0) <= 1e-9 * (1 + np.abs(b[fin]).max(initial=0)), (key, np.abs(a[fin] - b[fin]).max())
    ref = _oracle_solve(oracle, dev, batch=True, warm=(dev["wx"], dev["wy"]))
    assert np.array_equal(dev["code"], ref["code"]) and np.array_equal(dev["iter"], ref["iter"])
    assert np.array_equal(dev["x"], ref["x"]) and np.array_equal(dev["y"], ref["y"])
    dev3 = M.asif_swarm_device_step(st, ud, K, ticks=3)
    host3 = M.asif_swarm_step(B, K, ticks=3, seed=3)
    ref3 = _oracle_solve(oracle, dev3, batch=True, warm=(dev3["wx"], dev3["wy"]))          # the warm-started tick, too
    assert np.array_equal(dev3["iter"], ref3["iter"]) and np.array_equal(dev3["x"], ref3["x"]) and np.abs(dev3["wx"]).max() > 0
    assert np.array_equal(dev3["code"], host3["code"])
    assert np.abs(dev3["u"] - host3["u"]).max() <= 1e-6
    assert (dev3["code"] == 0).mean() > 0.9
