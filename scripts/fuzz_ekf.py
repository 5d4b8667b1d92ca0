"""Randomised parity sweep of the EKF entry points against the oracle: every dof, ny in 1..16 (per-lane and generic kernels),
predict / update / fused, shared or per-filter Q, R, dt, singular and indefinite innovation covariances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smooth_feedback_amd as sfb
from oracle import loader as O

def flat(M):
    return np.ascontiguousarray(M.transpose(0, 2, 1).reshape(M.shape[0], -1))

if __name__ == "__main__":
    if os.environ.get("KNOBS"):  # debug knobs of the library for this sweep: KNOBS="SFB_SP_GRID=4,SFB_SP_PAUSE=2" (sfb_debug_set)
        print("debug knobs:", sfb.debug_set_from(os.environ["KNOBS"]))
    N = int(os.environ.get("N", 300)); seed0 = int(os.environ.get("SEED", 1))
    bad = 0
    for it in range(N):
        rng = np.random.default_rng(2 * 10**6 + seed0 + it)
        dof = int(rng.integers(1, 17)); ny = int(rng.integers(1, 17)); B = int(rng.integers(1, 130))
        if rng.random() < 0.4: dof, ny = int(rng.choice([2, 3, 4, 6])), int(rng.choice([1, 2, 3]))
        G = rng.uniform(-1, 1, (B, dof, dof))
        P = flat(np.eye(dof)[None] + G @ G.transpose(0, 2, 1) / dof)
        A = flat(rng.uniform(-1, 1, (B, dof, dof)))
        Q = flat(0.1 * np.tile(np.eye(dof), (B, 1, 1)) + 0.01 * rng.uniform(-1, 1, (B, dof, dof)))
        dt = rng.uniform(0.01, 0.05, B)
        H = flat(rng.uniform(-1, 1, (B, ny, dof)))
        Gr = rng.uniform(-1, 1, (B, ny, ny))
        R = flat(0.1 * np.tile(np.eye(ny), (B, 1, 1)) + 0.01 * Gr @ Gr.transpose(0, 2, 1))
        kind = rng.random()
        if kind < 0.1: R[:] = 0.0; H[::2] = 0.0                 # singular S for every other filter
        elif kind < 0.2: R = -R                                  # indefinite S
        r = rng.uniform(-1, 1, (B, ny))
        sharedQ, sharedR, shareddt = rng.random() < 0.3, rng.random() < 0.3, rng.random() < 0.3
        Qa = Q[0].copy() if sharedQ else Q; Ra = R[0].copy() if sharedR else R; dta = float(dt[0]) if shareddt else dt
        mode = rng.integers(0, 3)
        ok = True
        if mode == 0:
            P1, _, _ = sfb.ekf_step_batch_host(P, dof, A=A, Q=Qa, dt=dta)
            ok = np.array_equal(P1, O.ekf_predict_batch(A, Qa, dta, P), equal_nan=True)
        elif mode == 1:
            P1, d1, i1 = sfb.ekf_step_batch_host(P, dof, H=H, R=Ra, r=r)
            ref, dref, iref = O.ekf_update_batch(H, Ra, r, P, dof)
            ok = np.array_equal(P1, ref, equal_nan=True) and np.array_equal(d1, dref, equal_nan=True) and np.array_equal(i1, iref)
        else:
            P1, d1, i1 = sfb.ekf_step_batch_host(P, dof, A=A, Q=Qa, dt=dta, H=H, R=Ra, r=r)
            ref, dref, iref = O.ekf_update_batch(H, Ra, r, O.ekf_predict_batch(A, Qa, dta, P), dof)
            ok = np.array_equal(P1, ref, equal_nan=True) and np.array_equal(d1, dref, equal_nan=True) and np.array_equal(i1, iref)
        if not ok:
            bad += 1
            print("MISMATCH seed", seed0 + it, "dof", dof, "ny", ny, "B", B, "mode", int(mode), "kind", kind, sharedQ, sharedR, shareddt)
            if bad >= 5: break
    print("fuzz ekf: %d configurations, %d mismatching" % (it + 1, bad))
