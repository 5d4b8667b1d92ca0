"""CPU checks of the device-side MPC assembly interface: the linearisation records of the C++ front
(MPC::fill_record) reproduce the host transcription (MPC::assemble, itself pinned against a numpy restatement
in test_mpc_host.py) through a numpy restatement of what the kernel does; sizes and argument errors of the
C-ABI.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from examples import models_lib as M


def assemble_from_records(L, rec, shared=None):
    """numpy restatement of mpc_assemble_kernel = ocp_to_qp_update_dyn / _cr / _ce (ocp_to_qp.hpp:240-373),
    one agent; the same operations in the same order (plain double arithmetic, no fused multiply-add)."""
    N, nx, nu, ncr, km, tf = L.N, L.nx, L.nu, L.ncr, L.kmesh, L.tf
    if shared is None:
        sz = [N * nx, N * nx, N * nx * nx, N * nx * nu, N * ncr, N * ncr * nx, N * ncr * nu, nx, nx * nx]
        f, dx, dfx, dfu, c, dcx, dcu, e, J = np.split(rec, np.cumsum(sz)[:-1])
    else:
        f, dx, c, e, J = np.split(rec, np.cumsum([N * nx, N * nx, N * ncr, nx])[:4])
        dfx, dfu, dcx, dcu = np.split(shared, np.cumsum([N * nx * nx, N * nx * nu, N * ncr * nx])[:3])
    f = f.reshape(N, nx); dx = dx.reshape(N, nx); dfx = dfx.reshape(N, nx, nx); dfu = dfu.reshape(N, nx, nu)
    c = c.reshape(N, ncr); dcx = dcx.reshape(N, ncr, nx); dcu = dcu.reshape(N, ncr, nu); J = J.reshape(nx, nx)
    A, lo, hi = [], [], []
    for node in range(N):
        s, i = divmod(node, km)
        s2 = f[node] + dx[node]
        adm = np.zeros((nx, nx)); off = 0
        for kind, dof in zip(L.kind, L.dof):
            if kind in (1, 2):
                a = s2[off:off + 3]
                adm[off + 0, off + 1] = -a[2]; adm[off + 0, off + 2] = a[1]
                adm[off + 1, off + 0] = a[2]; adm[off + 1, off + 2] = -a[0]
                if kind == 2:
                    adm[off + 2, off + 0] = -a[1]; adm[off + 2, off + 1] = a[0]
            off += dof
        for d in range(nx):
            for j in range(km + 1):
                dc = L.alpha[s] * L.D[j, i]
                if j == i:
                    for cc in range(nx):
                        v = 0.0 + tf * dfx[node, d, cc]
                        if len(L.kind):
                            v = v + (-tf / 2) * adm[d, cc]
                        if cc == d:
                            v = v - dc
                        A.append(v)
                else:
                    A.append(0.0 - dc)
            A.extend(0.0 + tf * dfu[node, d])
            lo.append(-tf * (f[node, d] - dx[node, d])); hi.append(lo[-1])
    for node in range(N):
        for d in range(ncr):
            A.extend(dcx[node, d]); A.extend(dcu[node, d])
            lo.append(L.crl[d] - c[node, d]); hi.append(L.cru[d] - c[node, d])
    A.extend(J.ravel())
    lo.extend(0.0 - e); hi.extend(0.0 - e)
    return np.array(A), np.array(lo), np.array(hi)


@pytest.mark.parametrize("variant,K", [(6, 10), (12, 50)])
def test_records_reproduce_the_host_transcription(variant, K):
    B = 3
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=4)
    L, rec = M.mpc_records(variant, K, B, seed=4)
    d = M.mpc_dims(variant, K)
    assert (L.n, L.m, L.nnzA) == (d["n"], d["m"], d["nnzA"])
    assert len(L.kind) == (2 if variant == 6 else 4)           # Bundle<SE2, R3> (x2)
    for b in range(B):
        A2, lo, hi = assemble_from_records(L, rec[b])
        assert np.array_equal(A2, Av[b]) and np.array_equal(lo, l[b]) and np.array_equal(hi, u[b])


def test_record_sizes_and_layout_errors(sfb):
    L = M.mpc_layout(12, 50)
    N = L.N
    assert L.record_doubles() == N * (2 * 12 + 144 + 24 + 2 + 24 + 4) + 12 + 144
    assert L.shared_jac_doubles == N * (144 + 24 + 24 + 4)
    assert L.record_doubles(True) == L.record_doubles() - L.shared_jac_doubles
    # the vehicle's Jacobians do not depend on the agent: the shared-Jacobian form carries the same information
    _, rec = M.mpc_records(12, 50, 2, seed=0)
    own, shared = L.split_shared(rec)
    assert own.shape[1] == L.record_doubles(True) and shared.shape[0] == L.shared_jac_doubles
    for b in range(2):
        for x, y in zip(assemble_from_records(L, rec[b]), assemble_from_records(L, own[b], shared)):
            assert np.array_equal(x, y)
    with pytest.raises(sfb._capi.SfbError):          # part dofs must sum to nx
        sfb.MPCLayout(6, 2, 2, 4, 3, 5.0, np.ones(3), np.zeros((5, 4)), parts=[(1, 3)], crl=[0, 0], cru=[1, 1]).record_doubles()
    with pytest.raises(sfb._capi.SfbError):          # beyond the kernel-argument tables
        sfb.MPCLayout(30, 2, 0, 4, 3, 5.0, np.ones(3), np.zeros((5, 4))).record_doubles()


def test_swarm_needs_a_matching_plan_and_a_device(sfb):
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(6, 10)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(6, 10))
    with pytest.raises(sfb._capi.SfbError) as e:      # layout of another horizon
        sfb.MPCSwarm(plan, M.mpc_layout(6, 30), Pv, np.zeros(d["n"]), 4)
    assert e.value.status == sfb._capi.SFB_ERR_INVALID_ARG
    if sfb._capi.device_count() == 0:
        with pytest.raises(sfb._capi.SfbError) as e:  # no CPU fallback
            sfb.MPCSwarm(plan, M.mpc_layout(6, 10), Pv, np.zeros(d["n"]), 4)
        assert e.value.status == sfb._capi.SFB_ERR_NO_DEVICE
