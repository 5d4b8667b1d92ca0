"""Generates tests/golden/mpc_sparse.npz and tests/golden/ekf.npz: inputs of the sparse MPC path and of the EKF
path with the CPU oracle's outputs (oracle/qp_sparse_oracle.c, oracle/ekf_oracle.c).
Run from the repo root:  python tests/golden/make_golden_sparse.py
The reference itself cannot be built in this environment (no Eigen / smooth / Boost), so these vectors pin
oracle <-> HIP kernel agreement and guard the oracle against regressions; the oracle <-> reference link is the
known-answer tests (tests/test_oracle_*.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import loader as O  # noqa: E402
import smooth_feedback_amd as sfb  # noqa: E402
from examples import models_lib as M  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def flat(Mx):  # (B, r, c) -> col-major flat (B, r*c)
    return np.ascontiguousarray(Mx.transpose(0, 2, 1).reshape(Mx.shape[0], -1))


def mpc():
    variant, K, B = 12, 50, 8   # BASELINE configs[2]: nx = 12, nu = 2, K = 50 -> n = m = 740
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3)
    keep = np.any(Av != 0.0, axis=0)
    stage = M.mpc_stage(variant, K)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=stage, keep=keep)
    perm, forder = plan.perm, plan.factor_order()
    Px, q = np.tile(Pv, (B, 1)), np.zeros((B, d["n"]))
    prm = O.default_params()   # MPCParams.qp{} defaults
    r = O.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l, u, perm=perm, forder=forder, params=prm, nthreads=8)
    l2, u2 = l + 1e-3 * (l == u), u + 1e-3 * (l == u)
    r2 = O.qp_sparse_solve_batch(Pp, Pi, Px, q, Ap, Aj, Av, l2, u2, perm=perm, forder=forder, params=prm, warm_x=r["x"],
                                 warm_y=r["y"], nthreads=8)
    out = dict(variant=variant, K=K, n=d["n"], m=d["m"], Pp=Pp, Pi=Pi, Pv=Pv, Ap=Ap, Aj=Aj, stage=stage, keep=keep,
               Av_kept=np.ascontiguousarray(Av[:, keep]), l=l, u=u, perm=perm, forder=forder)
    for tag, rr in (("cold", r), ("warm", r2)):
        for k in ("code", "iter", "x", "y", "obj"):
            out["%s_%s" % (tag, k)] = rr[k]
    print("mpc: codes", np.bincount(r["code"], minlength=7), "iters", r["iter"], "warm iters", r2["iter"], "nnzL", plan.nnzL)
    np.savez_compressed(os.path.join(HERE, "mpc_sparse.npz"), **out)


def ekf():
    rng = np.random.default_rng(2024)
    out = {}
    for tag, (dof, ny, B) in (("se2r3", (6, 3, 64)), ("generic", (10, 3, 16))):
        G = rng.uniform(-1, 1, (B, dof, dof))
        P0 = flat(np.eye(dof)[None] + G @ G.transpose(0, 2, 1) / dof)
        A = flat(rng.uniform(-1, 1, (B, dof, dof)))
        Q = flat(0.1 * np.tile(np.eye(dof), (B, 1, 1)) + 0.01 * rng.uniform(-1, 1, (B, dof, dof)))
        H = flat(rng.uniform(-1, 1, (B, ny, dof)))
        R = flat(0.1 * np.tile(np.eye(ny), (B, 1, 1)))
        r = rng.uniform(-1, 1, (3, B, ny))
        dt = rng.uniform(0.01, 0.05, B)
        P, Ps, ds = P0.copy(), [], []
        for tick in range(3):   # fused Euler predict + update, three consecutive ticks on the evolving covariance
            Pp = O.ekf_predict_batch(A, Q, dt, P)
            P, delta, info = O.ekf_update_batch(H, R, r[tick], Pp, dof)
            assert (info == 0).all()
            Ps.append(P.copy()); ds.append(delta.copy())
        Am, Ae = flat(rng.uniform(-1, 1, (B, dof, dof))), flat(rng.uniform(-1, 1, (B, dof, dof)))
        Prk = O.ekf_predict_batch(A, Q, dt, P0, stepper="rk4")
        Prk_tv = O.ekf_predict_batch(A, Q, dt, P0, stepper="rk4", A_mid=Am, A_end=Ae)
        out.update({tag + "_" + k: v for k, v in dict(dims=np.array([dof, ny, B]), P0=P0, A=A, A_mid=Am, A_end=Ae, Q=Q, H=H, R=R,
                                                      r=r, dt=dt, P_ticks=np.stack(Ps), delta_ticks=np.stack(ds), P_rk4=Prk,
                                                      P_rk4_tv=Prk_tv).items()})
    np.savez_compressed(os.path.join(HERE, "ekf.npz"), **out)
    print("ekf fixtures written")


if __name__ == "__main__":
    mpc()
    ekf()
