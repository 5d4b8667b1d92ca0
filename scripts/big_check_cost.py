import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import smooth_feedback_amd as sfb
rng = np.random.default_rng(5)
n, m = 3, 203
P = np.eye(n)[None].copy(); q = rng.uniform(-1, 1, (1, n))
A = np.zeros((1, n, m)); A[0, :, :200] = rng.uniform(-1, 1, (n, 200)); A[0, :, 200:] = np.eye(3)
A = np.ascontiguousarray(A)
l = np.full((1, m), -np.inf); l[0, :200] = -rng.uniform(0.5, 2, 200); l[0, 200:] = -1
u = np.full((1, m), np.inf); u[0, 200:] = 1
for sci in (25, 0, 5):
    for mi in (600, 1200):
        prm = sfb.QPSolverParams(max_iter=mi, polish=False, eps_abs=1e-30, eps_rel=1e-30, stop_check_iter=sci)
        sfb.solve_qp_batch_host(P.reshape(1, -1), q, A.reshape(1, -1), l, u, prm)
        t0 = time.perf_counter()
        for _ in range(5): r = sfb.solve_qp_batch_host(P.reshape(1, -1), q, A.reshape(1, -1), l, u, prm)
        dt = (time.perf_counter() - t0) / 5
        print("stop_check_iter", sci, "max_iter", mi, "%.3f ms" % (dt * 1e3), "iter", r.iter[0])
