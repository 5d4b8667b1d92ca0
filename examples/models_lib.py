"""ctypes loader of the example harness library (examples/models.cpp -> libsfb_models.so): concrete MPC / ASIF / EKF
models behind the C++ host front, used by bench.py, scripts/ and tests/ (`from examples import models_lib`)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "smooth_feedback_amd", "libsfb_models.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
        import smooth_feedback_amd  # noqa: F401  (loads libsfb.so / torch's HIP runtime first)
        L = C.CDLL(PATH)
        L.sfbx_lie_selftest.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mpc_dims(variant, K):
    v = [C.c_int() for _ in range(7)]
    assert lib().sfbx_mpc_dims(variant, K, *[C.byref(x) for x in v]) == 0
    return dict(zip(("n", "m", "nnzP", "nnzA", "Nx", "Nu", "N"), [x.value for x in v]))


def mpc_pattern(variant, K, tf=5.0):
    d = mpc_dims(variant, K)
    Pp = np.zeros(d["n"] + 1, np.int32); Pi = np.zeros(d["nnzP"], np.int32); Pv = np.zeros(d["nnzP"])
    Ap = np.zeros(d["m"] + 1, np.int32); Aj = np.zeros(d["nnzA"], np.int32)
    assert lib().sfbx_mpc_pattern(variant, K, C.c_double(tf), _p(Pp), _p(Pi), _p(Pv), _p(Ap), _p(Aj)) == 0
    return d, Pp, Pi, Pv, Ap, Aj


def mpc_assemble_batch(variant, K, batch, seed=0, tf=5.0, threads=8):
    d = mpc_dims(variant, K)
    Av = np.zeros((batch, d["nnzA"])); l = np.zeros((batch, d["m"])); u = np.zeros((batch, d["m"]))
    assert lib().sfbx_mpc_assemble_batch(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), _p(Av), _p(l),
                                         _p(u), threads) == 0
    return Av, l, u


def mesh(n_ivals, K):
    N = n_ivals * K
    nodes = np.zeros(N + 1); w = np.zeros(N + 1); D = np.zeros((K + 1) * K)
    assert lib().sfbx_mesh(n_ivals, K, _p(nodes), _p(w), _p(D)) == 0
    return nodes, w, D.reshape(K, K + 1).T  # D[j, i]


def mpc_stage(variant, K):
    d = mpc_dims(variant, K)
    st = np.zeros(d["n"] + d["m"], np.int32)
    assert lib().sfbx_mpc_stage(variant, K, _p(st)) == 0
    return st


def asif_basic_qp(x0, udes):
    """asif_to_qp of tests/test_asif.cpp:37-95 (host only): dict P (3,3), q, A (9,3), l, u."""
    x0 = np.asarray(x0, dtype=np.float64); udes = np.asarray(udes, dtype=np.float64)
    P = np.zeros(9); q = np.zeros(3); A = np.zeros(27); l = np.zeros(9); u = np.zeros(9)
    assert lib().sfbx_asif_basic_qp(_p(x0), _p(udes), _p(P), _p(q), _p(A), _p(l), _p(u)) == 0
    return dict(P=P.reshape(3, 3).T, q=q, A=A.reshape(3, 9).T, l=l, u=u)


def test_asif(which):
    """ASIFilter on the GPU (see models.h): dict u, code, iter, n, m and the QP (flat column-major) + x, y."""
    nmax, mmax = 8, 512
    u_out = np.zeros(3); code = C.c_int32(-1); it = C.c_uint32(0); dims = np.zeros(2, np.int32)
    P = np.zeros(nmax * nmax); q = np.zeros(nmax); A = np.zeros(mmax * nmax); l = np.zeros(mmax); u = np.zeros(mmax)
    x = np.zeros(nmax); y = np.zeros(mmax)
    rc = lib().sfbx_test_asif(which, _p(u_out), C.byref(code), C.byref(it), _p(dims), _p(P), _p(q), _p(A), _p(l), _p(u),
                              _p(x), _p(y))
    assert rc == 0
    n, m = int(dims[0]), int(dims[1])
    return dict(u=u_out, code=code.value, iter=it.value, n=n, m=m, P=P[:n * n].copy(), q=q[:n].copy(),
                A=A[:m * n].copy(), l=l[:m].copy(), ub=u[:m].copy(), x=x[:n].copy(), y=y[:m].copy())


def asif_swarm_step(batch, K, ticks=1, seed=0):
    n, m = 3, K + 3
    out = dict(u=np.zeros((batch, 2)), code=np.zeros(batch, np.int32), iter=np.zeros(batch, np.uint32),
               P=np.zeros((batch, n * n)), q=np.zeros((batch, n)), A=np.zeros((batch, m * n)), l=np.zeros((batch, m)),
               ub=np.zeros((batch, m)), x=np.zeros((batch, n)), y=np.zeros((batch, m)), wx=np.zeros((batch, n)),
               wy=np.zeros((batch, m)))
    rc = lib().sfbx_asif_swarm_step(C.c_int64(batch), C.c_uint64(seed), K, ticks, _p(out["u"]), _p(out["code"]),
                                    _p(out["iter"]), _p(out["P"]), _p(out["q"]), _p(out["A"]), _p(out["l"]), _p(out["ub"]),
                                    _p(out["x"]), _p(out["y"]), _p(out["wx"]), _p(out["wy"]))
    assert rc == 0
    return out


def mpc_layout(variant, K, tf=5.0):
    """MPCLayout (smooth_feedback_amd.mpc) of the variant's transcription, from the C++ front (MPC::device_layout)."""
    import smooth_feedback_amd as sfb
    dims = np.zeros(6, np.int32); alpha = np.zeros(128); D = np.zeros(72); kind = np.zeros(16, np.int32)
    dof = np.zeros(16, np.int32); crl = np.zeros(16); cru = np.zeros(16)
    assert lib().sfbx_mpc_layout(variant, K, C.c_double(tf), _p(dims), _p(alpha), _p(D), _p(kind), _p(dof), _p(crl), _p(cru)) == 0
    nx, nu, ncr, kmesh, nivals, nparts = [int(v) for v in dims]
    return sfb.MPCLayout(nx, nu, ncr, kmesh, nivals, tf, alpha[:nivals], D[:(kmesh + 1) * kmesh].reshape(kmesh + 1, kmesh),
                         parts=[(int(kind[g]), int(dof[g])) for g in range(nparts)], crl=crl[:ncr], cru=cru[:ncr])


def mpc_records(variant, K, batch, seed=0, tf=5.0, threads=8):
    """Linearisation records (MPC::fill_record) of the agents of mpc_assemble_batch."""
    L = mpc_layout(variant, K, tf)
    rec = np.zeros((batch, L.record_doubles()))
    assert lib().sfbx_mpc_records(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), _p(rec), threads) == 0
    return L, rec


def mpc_swarm_step(variant, K, batch, ticks, seed=1, tf=5.0, device=False):
    """`ticks` closed-loop ticks through MPCSwarm (host assembly) or MPCSwarmDevice; outputs of the last tick."""
    u0 = np.zeros((batch, 2)); codes = np.zeros(batch, np.int32); iters = np.zeros(batch, np.uint32)
    fn = lib().sfbx_mpc_swarm_device_step if device else lib().sfbx_mpc_swarm_step
    rc = fn(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), ticks, _p(u0), _p(codes), _p(iters))
    assert rc == 0, rc
    return u0, codes, iters


def mpc_swarm_step_multi(variant, K, batch, ticks, devices, seed=1, tf=5.0):
    """mpc_swarm_step with every batched solve sharded over `devices` from this one process (MPCSwarm +
    QPSolver::shard_over_devices + sfb_set_devices)."""
    u0 = np.zeros((batch, 2)); codes = np.zeros(batch, np.int32); iters = np.zeros(batch, np.uint32)
    dev = (C.c_int * len(devices))(*devices)
    rc = lib().sfbx_mpc_swarm_step_multi(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), ticks, dev, len(devices),
                                         _p(u0), _p(codes), _p(iters))
    assert rc == 0, rc
    return u0, codes, iters


def last_tick_seconds(n):
    """wall seconds of every swarm.step() of the last mpc_swarm_step call"""
    out = np.zeros(n)
    k = lib().sfbx_last_tick_seconds(_p(out), n)
    return out[:k]


def mpc_doubleintegrator(ticks):
    """examples/mpc_doubleintegrator.cpp in closed loop, with the factor reuse of the solver front and without."""
    u = np.zeros(ticks); it = np.zeros(ticks, np.uint32); codes = np.zeros(ticks, np.int32)
    ur = np.zeros(ticks); itr = np.zeros(ticks, np.uint32); cnt = C.c_int64(0); sec = np.zeros(2)
    rc = lib().sfbx_test_mpc_doubleintegrator(ticks, _p(u), _p(it), _p(codes), _p(ur), _p(itr), C.byref(cnt), _p(sec))
    assert rc == 0, rc
    return dict(u=u, iter=it, code=codes, u_ref=ur, iter_ref=itr, reuse_count=cnt.value, seconds=sec)


def ocp_to_qp_basic(solve=False):
    """tests/test_ocp_to_qp.cpp:41-107 with the generic ocp_to_qp() front; solve=True also runs solve_qp and
    qpsol_to_ocpsol (needs a GPU)."""
    out = np.full(18, np.nan)
    rc = lib().sfbx_test_ocp_to_qp_basic(_p(out), int(solve))
    assert rc == 0, rc
    return out


_dev = None


def dev_lib():
    """libsfb_models_dev.so: the device-side fronts (examples/models_device.hip, compiled by hipcc)."""
    global _dev
    if _dev is None:
        lib()  # libsfb.so / the host harness first
        _dev = C.CDLL(os.path.join(ROOT, "smooth_feedback_amd", "libsfb_models_dev.so"))
    return _dev


def asif_swarm_states(batch, seed=0):
    st = np.zeros((batch, 7)); ud = np.zeros((batch, 2))
    assert lib().sfbx_asif_swarm_states(C.c_int64(batch), C.c_uint64(seed), _p(st), _p(ud)) == 0
    return st, ud


def asif_swarm_device_step(states, udes, K, ticks=1):
    """ASIFSwarmDevice (assembly and solve on the GPU) from the states of asif_swarm_states; outputs like asif_swarm_step."""
    batch = len(states)
    n, m = 3, K + 3
    out = dict(u=np.zeros((batch, 2)), code=np.zeros(batch, np.int32), iter=np.zeros(batch, np.uint32), P=np.zeros((batch, n * n)),
               q=np.zeros((batch, n)), A=np.zeros((batch, m * n)), l=np.zeros((batch, m)), ub=np.zeros((batch, m)),
               x=np.zeros((batch, n)), y=np.zeros((batch, m)), wx=np.zeros((batch, n)), wy=np.zeros((batch, m)), seconds=np.zeros(ticks))
    st = np.ascontiguousarray(states, dtype=np.float64); ud = np.ascontiguousarray(udes, dtype=np.float64)
    rc = dev_lib().sfbx_asif_swarm_device_step(C.c_int64(batch), K, ticks, _p(st), _p(ud), _p(out["u"]), _p(out["code"]), _p(out["iter"]),
                                               _p(out["P"]), _p(out["q"]), _p(out["A"]), _p(out["l"]), _p(out["ub"]), _p(out["x"]),
                                               _p(out["y"]), _p(out["wx"]), _p(out["wy"]), _p(out["seconds"]))
    assert rc == 0, rc
    return out


def mpc_swarm_devlin_step(variant, K, batch, ticks, seed=1, tf=5.0, probe_empty=False, want_records=True):
    """MPCSwarmDeviceLin (linearisation, assembly and solve on the GPU): the agents and closed loop of mpc_swarm_step;
    also the records of the last tick as the device wrote them."""
    dims = mpc_dims(variant, K)
    Nx, Nu, N = dims["Nx"], dims["Nu"], dims["N"]
    full = N * (2 * Nx + Nx * Nx + Nx * Nu + 2 + 2 * Nx + 2 * Nu) + Nx + Nx * Nx
    out = dict(u0=np.zeros((batch, 2)), code=np.zeros(batch, np.int32), iter=np.zeros(batch, np.uint32), seconds=np.zeros(ticks))
    rec = np.zeros((batch, full)) if want_records else None
    rd = C.c_int64(0); packed = C.c_int32(0)
    rc = dev_lib().sfbx_mpc_swarm_devlin_step(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), ticks, int(probe_empty),
                                              _p(out["u0"]), _p(out["code"]), _p(out["iter"]), _p(rec) if want_records else None,
                                              C.byref(rd), C.byref(packed), _p(out["seconds"]))
    assert rc == 0, rc
    out["record_doubles"] = rd.value
    out["packed"] = bool(packed.value)
    if want_records:
        out["records"] = rec.reshape(-1)[: batch * rd.value].reshape(batch, rd.value).copy()
    return out


def mpc_swarm_devlin_step_multi(variant, K, batch, ticks, devices, seed=1, tf=5.0, thread_per_shard=False):
    """MPCSwarmMultiDeviceLin (multi_device.hpp): mpc_swarm_devlin_step with the agents sharded over `devices` from this one
    process, one resident swarm per shard."""
    out = dict(u0=np.zeros((batch, 2)), code=np.zeros(batch, np.int32), iter=np.zeros(batch, np.uint32), seconds=np.zeros(ticks))
    dev = (C.c_int * len(devices))(*devices)
    rc = dev_lib().sfbx_mpc_swarm_devlin_step_multi(variant, K, C.c_double(tf), C.c_int64(batch), C.c_uint64(seed), ticks, dev, len(devices),
                                                    int(thread_per_shard), _p(out["u0"]), _p(out["code"]), _p(out["iter"]), _p(out["seconds"]))
    assert rc == 0, rc
    return out


def ekf_swarm_device_multi(states, P0, y, devices, tau=0.1, dt=0.0, rk4=False, fused=False, thread_per_shard=False):
    """EKFSwarmMultiDevice (multi_device.hpp): ekf_swarm_device with the filters sharded over `devices`."""
    batch, steps = len(states), len(y)
    out = dict(states=np.zeros((batch, 7)), P=np.zeros((batch, 36)), info=np.zeros(batch, np.int32))
    st = np.ascontiguousarray(states, dtype=np.float64); P0 = np.ascontiguousarray(P0, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    dev = (C.c_int * len(devices))(*devices)
    rc = dev_lib().sfbx_ekf_swarm_device_multi(C.c_int64(batch), steps, int(rk4), int(fused), C.c_double(tau), C.c_double(dt), dev, len(devices),
                                               int(thread_per_shard), _p(st), _p(P0), _p(y), _p(out["states"]), _p(out["P"]), _p(out["info"]))
    assert rc == 0, rc
    return out


def ekf_swarm_inputs(batch, steps, seed=0):
    """states [batch][7] of asif_swarm_states, SPD covariances, measurements near the states' (x, y, v0)"""
    st, _ = asif_swarm_states(batch, seed)
    rng = np.random.default_rng(seed)
    G = rng.uniform(-1, 1, (batch, 6, 6))
    P0 = (np.eye(6)[None] * 0.5 + G @ G.transpose(0, 2, 1) / 12).reshape(batch, 36)
    y = st[None, :, [0, 1, 4]] + rng.normal(0, 0.2, (steps, batch, 3))
    return st, np.ascontiguousarray(P0), np.ascontiguousarray(y)


def ekf_swarm_device(states, P0, y, tau=0.1, dt=0.0, rk4=False, fused=False):
    """EKFSwarmDevice (ekf_device.hpp): len(y) predict+update rounds of every filter on the GPU."""
    batch, steps = len(states), len(y)
    out = dict(states=np.zeros((batch, 7)), P=np.zeros((batch, 36)), info=np.zeros(batch, np.int32), seconds=np.zeros(steps))
    st = np.ascontiguousarray(states, dtype=np.float64); P0 = np.ascontiguousarray(P0, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    rc = dev_lib().sfbx_ekf_swarm_device(C.c_int64(batch), steps, int(rk4), int(fused), C.c_double(tau), C.c_double(dt), _p(st), _p(P0),
                                         _p(y), _p(out["states"]), _p(out["P"]), _p(out["info"]), _p(out["seconds"]))
    assert rc == 0, rc
    return out


def ekf_swarm_host(states, P0, y, tau=0.1, dt=0.0, rk4=False):
    """the same through one host EKF<> object per filter"""
    batch, steps = len(states), len(y)
    out = dict(states=np.zeros((batch, 7)), P=np.zeros((batch, 36)))
    st = np.ascontiguousarray(states, dtype=np.float64); P0 = np.ascontiguousarray(P0, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
    rc = lib().sfbx_ekf_swarm_host(C.c_int64(batch), steps, int(rk4), C.c_double(tau), C.c_double(dt), _p(st), _p(P0), _p(y),
                                   _p(out["states"]), _p(out["P"]))
    assert rc == 0, rc
    return out


def vehicle_swarm_sim(batch, ticks, K_mpc=30, K_asif=200, seed=0):
    """examples/mpc_asif_vehicle.cpp's closed loop for a swarm, MPC and ASI filter on the GPU (models_device.hip)."""
    out = dict(xy=np.zeros((ticks, batch, 2)), u_mpc=np.zeros((ticks, batch, 2)), u_asif=np.zeros((ticks, batch, 2)),
               mpc_bad=np.zeros(ticks, np.int32), asif_bad=np.zeros(ticks, np.int32), hmin=np.zeros(ticks), seconds=np.zeros((ticks, 2)))
    rc = dev_lib().sfbx_vehicle_swarm_sim(C.c_int64(batch), K_mpc, K_asif, ticks, C.c_uint64(seed), _p(out["xy"]), _p(out["u_mpc"]),
                                          _p(out["u_asif"]), _p(out["mpc_bad"]), _p(out["asif_bad"]), _p(out["hmin"]), _p(out["seconds"]))
    assert rc == 0, rc
    return out
