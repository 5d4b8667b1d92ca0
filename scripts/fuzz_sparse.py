"""Randomised parity sweep of the shared-pattern sparse path against the sparse oracle: random patterns (empty rows /
columns, upper-only or full P), plain and pruned plans (with items that violate the mask), solver parameters, warm starts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import smooth_feedback_amd as sfb
from oracle import loader as O
from sparse_cases import dense_batch_to_sparse
from fuzz_dense import oparams  # noqa: E402

if __name__ == "__main__":
    if os.environ.get("KNOBS"):  # debug knobs of the library for this sweep: KNOBS="SFB_SP_GRID=4,SFB_SP_PAUSE=2" (sfb_debug_set)
        print("debug knobs:", sfb.debug_set_from(os.environ["KNOBS"]))
    N = int(os.environ.get("N", 200)); seed0 = int(os.environ.get("SEED", 1))
    bad = 0
    for it in range(N):
        rng = np.random.default_rng(10**6 + seed0 + it)
        n = int(rng.integers(2, 40)); m = int(rng.integers(2, 60)); B = int(rng.integers(1, int(os.environ.get("BMAX", 24))))
        P, q, A, l, u = sfb.random_qp_batch(int(rng.integers(1, 10**6)), B, m, n, float(rng.choice([0.05, 0.2, 0.6])))
        Pp, Pi, Px, Ap, Aj, Ax = dense_batch_to_sparse(P, A, n, m, upper_only=bool(rng.random() < 0.5))
        mask = rng.random((B, m))
        l = np.where(mask < 0.15, -np.inf, l); u = np.where((mask > 0.15) & (mask < 0.3), np.inf, u); l = np.where(mask > 0.9, u, l)
        prm = sfb.QPSolverParams(alpha=float(rng.choice([1.0, 1.6])), rho=float(rng.choice([0.01, 0.1, 1.0])), scaling=bool(rng.random() < 0.7),
                                 eps_abs=float(rng.choice([1e-3, 1e-6])), eps_rel=float(rng.choice([1e-3, 1e-6])),
                                 max_iter=int(rng.choice([0, 1, 26, 27, 60, 400])), stop_check_iter=int(rng.choice([1, 2, 5, 25])),
                                 polish=bool(rng.random() < 0.7), polish_iter=int(rng.choice([0, 1, 5])))
        keep = None
        if rng.random() < 0.5 and Ax.shape[1] > 4:   # pruned plan: declare a random third of A's entries zero, zero them in most items
            keep = (rng.random(Ax.shape[1]) > 0.33)
            viol = rng.random(B) < 0.2
            Ax = np.where(keep[None, :] | viol[:, None], Ax, 0.0)
        plan = sfb.SparseQPPlan(n, m, Pp, Pi, Ap, Aj, ordering=int(rng.integers(0, 2)), keep=keep)
        warm = rng.random() < 0.3
        wx = rng.uniform(-1, 1, (B, n)) if warm else None; wy = rng.uniform(-1, 1, (B, m)) if warm else None
        r = plan.solve_batch_host(Px, q, Ax, l, u, prm, warm_x=wx, warm_y=wy)
        ok = True
        groups = [(np.ones(B, bool), plan.factor_order())]
        if keep is not None:
            isbad = np.any((Ax != 0) & ~keep[None, :], axis=1)
            groups = [(~isbad, plan.factor_order()), (isbad, plan.factor_order(fallback=True))]
        for sel, fo in groups:
            if not sel.any(): continue
            ref = O.qp_sparse_solve_batch(Pp, Pi, Px[sel], q[sel], Ap, Aj, Ax[sel], l[sel], u[sel], perm=plan.perm, forder=fo, params=oparams(prm),
                                          warm_x=None if wx is None else wx[sel], warm_y=None if wy is None else wy[sel], nthreads=8)
            ok = ok and (np.array_equal(r.code[sel], ref["code"]) and np.array_equal(r.iter[sel], ref["iter"])
                         and np.array_equal(r.primal[sel], ref["x"], equal_nan=True) and np.array_equal(r.dual[sel], ref["y"], equal_nan=True))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed0 + it, "n", n, "m", m, "B", B, "pruned", keep is not None, prm, "warm", warm)
            if bad >= 5: break
    print("fuzz sparse: %d configurations, %d mismatching" % (it + 1, bad))
