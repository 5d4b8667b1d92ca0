"""Device-side MPC assembly and the device-resident swarm (sfb_mpc_* in include/sfb.h): thin ctypes
mirror used by tests, scripts and bench.py.  The linearisation records come from the host front
(MPC::fill_record, include/smooth_feedback_amd/mpc.hpp)."""
import ctypes as C
from typing import Optional

import numpy as np

from . import _capi
from .qp import QPSolverParams, SparseQPPlan, _f64, _ptr

LIE_RN, LIE_SE2, LIE_SO3 = 0, 1, 2


class MPCLayout:
    """sfb_mpc_layout: transcription of an MPC problem on an LGR mesh (ocp_to_qp_allocate, ocp_to_qp.hpp:40-114).
    parts: [(kind, dof), ...] components of the state bundle, empty for a commutative state."""

    def __init__(self, nx, nu, ncr, kmesh, nivals, tf, alpha, D, parts=(), crl=(), cru=(), jac_keep=None):
        self.nx, self.nu, self.ncr, self.kmesh, self.nivals, self.tf = int(nx), int(nu), int(ncr), int(kmesh), int(nivals), float(tf)
        self.alpha = _f64(alpha, (self.nivals,))
        self.D = _f64(D, (self.kmesh + 1, self.kmesh))          # D[j, i]
        self.kind = np.ascontiguousarray([p[0] for p in parts], dtype=np.int32)
        self.dof = np.ascontiguousarray([p[1] for p in parts], dtype=np.int32)
        self.crl = _f64(crl, (self.ncr,))
        self.cru = _f64(cru, (self.ncr,))
        self.N = self.kmesh * self.nivals
        self.n = self.nx * (self.N + 1) + self.nu * self.N
        self.m = self.N * self.nx + self.N * self.ncr + self.nx
        # jac_keep: flags over [dfdx nx*nx | dfdu nx*nu | dcdx ncr*nx | dcdu ncr*nu | J nx*nx] (packed records, sfb.h)
        self.jac_keep = None if jac_keep is None else np.ascontiguousarray(jac_keep, dtype=np.uint8).reshape(-1)
        if self.jac_keep is not None and self.jac_keep.size != self.jac_flags:
            raise ValueError("jac_keep must have %d flags" % self.jac_flags)
        self.c = _capi.SfbMPCLayout(
            self.nx, self.nu, self.ncr, self.kmesh, self.nivals, self.tf, _ptr(self.alpha), _ptr(self.D), len(self.kind),
            _ptr(self.kind) if len(self.kind) else None, _ptr(self.dof) if len(self.dof) else None,
            _ptr(self.crl) if self.ncr else None, _ptr(self.cru) if self.ncr else None,
            _ptr(self.jac_keep) if self.jac_keep is not None else None)

    @property
    def jac_flags(self):
        return 2 * self.nx * self.nx + self.nx * self.nu + self.ncr * self.nx + self.ncr * self.nu

    def _full_blocks(self, records):
        N, nx, nu, ncr = self.N, self.nx, self.nu, self.ncr
        sz = [N * nx, N * nx, N * nx * nx, N * nx * nu, N * ncr, N * ncr * nx, N * ncr * nu, nx, nx * nx]
        off = np.concatenate([[0], np.cumsum(sz)])
        return [records[:, off[i]:off[i + 1]] for i in range(9)]

    def jac_keep_of(self, records):
        """Flags of the Jacobian entries that are non-zero in some node of some of the given FULL records."""
        N, nx, nu, ncr = self.N, self.nx, self.nu, self.ncr
        f, dx, dfx, dfu, c, dcx, dcu, e, J = self._full_blocks(np.asarray(records))
        B = len(f)
        parts = [np.any(dfx.reshape(B * N, nx * nx) != 0, axis=0), np.any(dfu.reshape(B * N, nx * nu) != 0, axis=0),
                 np.any(dcx.reshape(B * N, ncr * nx) != 0, axis=0), np.any(dcu.reshape(B * N, ncr * nu) != 0, axis=0),
                 np.any(J.reshape(B, nx * nx) != 0, axis=0)]
        return np.concatenate(parts).astype(np.uint8)

    def pack_records(self, records):
        """[batch][full record] -> [batch][packed record] for this layout's jac_keep (the masked entries must be zero)."""
        if self.jac_keep is None:
            return np.ascontiguousarray(records)
        N, nx, nu, ncr = self.N, self.nx, self.nu, self.ncr
        f, dx, dfx, dfu, c, dcx, dcu, e, J = self._full_blocks(np.asarray(records))
        B = len(f)
        k = self.jac_keep.astype(bool)
        o = np.cumsum([0, nx * nx, nx * nu, ncr * nx, ncr * nu, nx * nx])
        kfx, kfu, kcx, kcu, kJ = [k[o[i]:o[i + 1]] for i in range(5)]
        for blk, kk, per in ((dfx, kfx, nx * nx), (dfu, kfu, nx * nu), (dcx, kcx, ncr * nx), (dcu, kcu, ncr * nu), (J, kJ, nx * nx)):
            if np.any(blk.reshape(B, -1, per)[:, :, ~kk] != 0):
                raise ValueError("a Jacobian entry outside jac_keep is not zero")
        pk = lambda blk, kk, per: blk.reshape(B, -1, per)[:, :, kk].reshape(B, -1)
        return np.ascontiguousarray(np.hstack([f, dx, pk(dfx, kfx, nx * nx), pk(dfu, kfu, nx * nu), c, pk(dcx, kcx, ncr * nx),
                                               pk(dcu, kcu, ncr * nu), e, pk(J, kJ, nx * nx)]))

    def record_doubles(self, shared_jac=False):
        v = _capi.lib.sfb_mpc_record_doubles(C.byref(self.c), int(bool(shared_jac)))
        if v < 0:
            raise _capi.SfbError(_capi.SFB_ERR_INVALID_ARG, _capi.lib.sfb_last_error().decode())
        return v

    @property
    def shared_jac_doubles(self):
        return _capi.lib.sfb_mpc_shared_jac_doubles(C.byref(self.c))

    @property
    def nnzA(self):
        return _capi.lib.sfb_mpc_nnzA(C.byref(self.c))

    def split_shared(self, records):
        """[batch][full record] -> ([batch][record without Jacobians], Jacobian record of agent 0)."""
        f, dx, dfx, dfu, c, dcx, dcu, e, J = self._full_blocks(records)
        return (np.ascontiguousarray(np.hstack([f, dx, c, e, J])),
                np.ascontiguousarray(np.hstack([dfx[0], dfu[0], dcx[0], dcu[0]])))

    def assemble_batch_device(self, batch, d_records, d_Ax, d_l, d_u, d_shared_jac=0, stream=0):
        """sfb_mpc_assemble_batch on device pointers (ints); asynchronous on `stream`."""
        _capi.check(_capi.lib.sfb_mpc_assemble_batch(C.byref(self.c), batch, d_records, d_shared_jac or None, d_Ax, d_l,
                                                     d_u, stream or None))


class MPCSwarm:
    """sfb_mpc_swarm: `agents` controllers with one transcription, resident on the device."""

    def __init__(self, plan: SparseQPPlan, layout: MPCLayout, Px, q, agents):
        self.plan, self.layout, self.agents = plan, layout, int(agents)
        Px = _f64(Px, (plan.nnzP,)); q = _f64(q, (plan.n,))
        h = C.c_void_p()
        _capi.check(_capi.lib.sfb_mpc_swarm_create(plan._h, C.byref(layout.c), _ptr(Px), _ptr(q), self.agents, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib.sfb_mpc_swarm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_warmstart(self):
        _capi.check(_capi.lib.sfb_mpc_swarm_reset_warmstart(self._h))

    def set_jac_keep(self, jac_keep):
        """sfb_mpc_swarm_set_jac_keep: switch the record packing (None = unpacked); returns the new record length."""
        v = C.c_int64()
        k = None if jac_keep is None else np.ascontiguousarray(jac_keep, dtype=np.uint8)
        _capi.check(_capi.lib.sfb_mpc_swarm_set_jac_keep(self._h, _ptr(k), C.byref(v)))
        self._rec_doubles = v.value
        return v.value

    def step_host(self, records, prm: Optional[QPSolverParams] = None, shared_jac=None, warmstart=True, full=False):
        """One tick.  Returns (du0 [agents, nu], code, iter[, primal, dual])."""
        B, L = self.agents, self.layout
        rd = getattr(self, "_rec_doubles", None) if shared_jac is None else None
        records = _f64(records, (B, rd if rd is not None else L.record_doubles(shared_jac is not None)))
        if shared_jac is not None:
            shared_jac = _f64(shared_jac, (L.shared_jac_doubles,))
        du0 = np.empty((B, L.nu)); it = np.empty(B, dtype=np.uint32); code = np.empty(B, dtype=np.int32)
        x = np.empty((B, L.n)) if full else None
        y = np.empty((B, L.m)) if full else None
        cp = (prm or QPSolverParams()).to_c()
        _capi.check(_capi.lib.sfb_mpc_swarm_step_host(self._h, C.byref(cp), _ptr(records), _ptr(shared_jac), int(bool(warmstart)),
                                                      _ptr(du0), _ptr(it), _ptr(code), _ptr(x), _ptr(y)))
        return (du0, code, it, x, y) if full else (du0, code, it)

    def debug_buffers(self):
        a, l, u = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _capi.check(_capi.lib.sfb_mpc_swarm_debug_buffers(self._h, C.byref(a), C.byref(l), C.byref(u)))
        return a.value, l.value, u.value
