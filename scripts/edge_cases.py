import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import smooth_feedback_amd as sfb
from oracle import loader as O
prm = sfb.QPSolverParams(max_iter=200)
for (n, m, B) in ((32, 32, 5), (32, 33, 5), (3, 61, 3), (3, 62, 3), (500, 524, 2), (500, 525, 1)):
    P, q, A, l, u = sfb.random_qp_batch(1, B, m, n, 0.3)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    ref = O.qp_dense_solve_batch(P, q, A, l, u, n=n, m=m, params=O.params_from(prm) if hasattr(O, "params_from") else None) if False else None
    print(n, m, "k", n + m, "codes", r.code, "iters", r.iter)
# empty batch
r = sfb.solve_qp_batch_host(np.zeros((0, 4)), np.zeros((0, 2)), np.zeros((0, 6)), np.zeros((0, 3)), np.zeros((0, 3)), prm)
print("empty batch ok", r.code.shape)
# NaN input: reported per item, other items unaffected
P, q, A, l, u = sfb.random_qp_batch(2, 4, 20, 10, 1.0)
q2 = q.copy(); q2[1, 0] = np.nan
r1 = sfb.solve_qp_batch_host(P, q, A, l, u, prm); r2 = sfb.solve_qp_batch_host(P, q2, A, l, u, prm)
print("nan item code", r2.code, "others equal", np.array_equal(r1.primal[[0, 2, 3]], r2.primal[[0, 2, 3]]))
