import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import smooth_feedback_amd as sfb
for n, m in ((40, 60), (64, 64)):
    P, q, A, l, u = sfb.random_qp_batch(5, 1, m, n, 1.0)
    prm = sfb.QPSolverParams(max_iter=200, polish=False, eps_abs=1e-30, eps_rel=1e-30)
    r = sfb.solve_qp_batch_host(P, q, A, l, u, prm)
    print(n, m, r.iter)
