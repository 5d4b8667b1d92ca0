import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import smooth_feedback_amd as sfb
for n, m in ((40, 60), (30, 40), (50, 78), (60, 90), (100, 156)):
    P, q, A, l, u = sfb.random_qp_batch(11, 4, m, n, 0.6)
    u = u + 5.0
    prm = sfb.QPSolverParams(max_iter=2000)
    sfb.solve_qp_batch_host(P[:1], q[:1], A[:1], l[:1], u[:1], prm)
    t0 = time.perf_counter()
    for _ in range(3): r1 = sfb.solve_qp_batch_host(P[:1], q[:1], A[:1], l[:1], u[:1], prm)
    t1 = (time.perf_counter() - t0) / 3
    print("n=%d m=%d: one QP %.2f ms (iter %d) -> %.1f us per iteration incl. setup" % (n, m, t1 * 1e3, r1.iter[0], t1 * 1e6 / max(1, r1.iter[0])))
