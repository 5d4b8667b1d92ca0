"""Randomised parity sweep of the dense entry point against the dense oracle: sizes, densities, solver parameters, infinite
and equal bounds, warm starts.  Prints the first mismatch (seed + configuration) or a summary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smooth_feedback_amd as sfb
from oracle import loader as O

def oparams(prm):
    return O.default_params(alpha=prm.alpha, rho=prm.rho, sigma=prm.sigma, scaling=int(prm.scaling), eps_abs=prm.eps_abs,
                            eps_rel=prm.eps_rel, eps_primal_inf=prm.eps_primal_inf, eps_dual_inf=prm.eps_dual_inf,
                            max_iter=-1 if prm.max_iter is None else prm.max_iter, stop_check_iter=prm.stop_check_iter,
                            polish=int(prm.polish), polish_iter=prm.polish_iter, delta=prm.delta)

def main():
  N = int(os.environ.get("N", 300)); seed0 = int(os.environ.get("SEED", 1))
  bad = 0
  for it in range(N):
      rng = np.random.default_rng(seed0 + it)
      n = int(rng.integers(1, 14)); m = int(rng.integers(1, 24))
      if rng.random() < 0.15: n, m = int(rng.integers(20, 30)), int(rng.integers(20, 36))      # one-per-wave sizes
      if rng.random() < 0.05: n, m = int(rng.integers(3, 40)), int(rng.integers(62, 90))       # big kernel
      if os.environ.get("BIG") == "1":                                                           # big kernel only: filter shapes, block-structured A
          n, m = int(rng.integers(1, 6)), int(rng.integers(64, 330))
          if rng.random() < 0.3: n, m = int(rng.integers(6, 60)), int(rng.integers(60, 200))
      if os.environ.get("MID") == "1":                                                           # the on-chip block-sweep kernel: 32 < n + m <= 128, every shape
          kk = int(rng.integers(33, 129)); n = int(rng.integers(1, kk)); m = kk - n
          if rng.random() < 0.3: n, m = int(rng.integers(8, 50)), int(rng.integers(25, 79))
      B = int(rng.integers(1, 40 if os.environ.get("BIG") != "1" else 6))
      P, q, A, l, u = sfb.random_qp_batch(int(rng.integers(1, 10**6)), B, m, n, float(rng.choice([0.1, 0.5, 1.0])))
      if os.environ.get("BIG") == "1" and rng.random() < 0.5:   # rows that touch one variable only / empty rows: long chain-free runs
          Am = A.reshape(B, n, m).copy()
          keepv = rng.integers(0, n, m)
          for j in range(n): Am[:, j, :] *= (keepv == j) | (rng.random(m) < 0.2)
          A = np.ascontiguousarray(Am.reshape(B, -1))
      mask = rng.random((B, m))
      l = np.where(mask < 0.15, -np.inf, l); u = np.where((mask > 0.15) & (mask < 0.3), np.inf, u)
      l = np.where(mask > 0.9, u, l)                       # equalities
      both = rng.random((B, m)) < 0.05
      l = np.where(both, -np.inf, l); u = np.where(both, np.inf, u)
      prm = sfb.QPSolverParams(alpha=float(rng.choice([1.0, 1.6, 1.8])), rho=float(rng.choice([0.01, 0.1, 1.0])),
                               sigma=float(rng.choice([1e-6, 1e-3])), scaling=bool(rng.random() < 0.7),
                               eps_abs=float(rng.choice([1e-3, 1e-6])), eps_rel=float(rng.choice([1e-3, 1e-6, 0.0])),
                               max_iter=int(rng.choice([0, 1, 2, 26, 27, 60, 400])), stop_check_iter=int(rng.choice([1, 2, 5, 25, 0])),
                               polish=bool(rng.random() < 0.7), polish_iter=int(rng.choice([0, 1, 5])), delta=float(rng.choice([1e-6, 1e-4])))
      warm = rng.random() < 0.3
      wx = rng.uniform(-1, 1, (B, n)) if warm else None; wy = rng.uniform(-1, 1, (B, m)) if warm else None
      r = sfb.solve_qp_batch_host(P, q, A, l, u, prm, warm_x=wx, warm_y=wy)
      ref = O.qp_dense_solve_batch(P, q, A, l, u, params=oparams(prm), warm_x=wx, warm_y=wy, nthreads=8)
      ok = (np.array_equal(r.code, ref["code"]) and np.array_equal(r.iter, ref["iter"]) and np.array_equal(r.primal, ref["x"], equal_nan=True)
            and np.array_equal(r.dual, ref["y"], equal_nan=True) and np.array_equal(r.objective, ref["obj"], equal_nan=True))
      if not ok:
          bad += 1
          w = np.nonzero((r.code != ref["code"]) | (r.iter != ref["iter"]) | ~np.all((r.primal == ref["x"]) | (np.isnan(r.primal) & np.isnan(ref["x"])), axis=1))[0]
          print("MISMATCH seed", seed0 + it, "n", n, "m", m, "B", B, prm, "warm", warm, "items", w[:5], "codes", r.code[w[:5]], ref["code"][w[:5]],
                "iters", r.iter[w[:5]], ref["iter"][w[:5]])
          if bad >= 5: break
  print("fuzz: %d configurations, %d mismatching" % (it + 1, bad))


if __name__ == "__main__":
    if os.environ.get("KNOBS"):  # debug knobs of the library for this sweep: KNOBS="SFB_SP_GRID=4,SFB_SP_PAUSE=2" (sfb_debug_set)
        print("debug knobs:", sfb.debug_set_from(os.environ["KNOBS"]))
    main()
