import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import smooth_feedback_amd as sfb
rng = np.random.default_rng(5)
n, m = 3, 203
# an ASIF-shaped QP: 3 variables, 200 barrier rows + 3 box rows
P = np.eye(n)[None].copy(); q = rng.uniform(-1, 1, (1, n))
A = np.zeros((1, n, m)); A[0, :, :200] = rng.uniform(-1, 1, (n, 200)); A[0, :, 200:] = np.eye(3)
A = np.ascontiguousarray(A)  # column-major (m x n) as [n][m]
l = np.full((1, m), -np.inf); l[0, :200] = -rng.uniform(0.5, 2, 200); l[0, 200:] = -1
u = np.full((1, m), np.inf); u[0, 200:] = 1
prm = sfb.QPSolverParams(max_iter=600, polish=False, eps_abs=1e-30, eps_rel=1e-30)
r = sfb.solve_qp_batch_host(P.reshape(1, -1), q, A.reshape(1, -1), l, u, prm)
print(r[3] if isinstance(r, tuple) else getattr(r, "iter", r))
