import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import smooth_feedback_amd as sfb
from examples import models_lib as M
for which in (1, 0):
    r = M.test_asif(which)
    t0 = time.perf_counter(); r = M.test_asif(which); dt = time.perf_counter() - t0
    n, m = r["n"], r["m"]
    print("asif case", which, "n", n, "m", m, "iter", r["iter"], "code", r["code"], "whole filter call %.2f ms" % (dt * 1e3))
    P = r["P"][None]; q = r["q"][None]; A = r["A"][None]; l = r["l"][None]; u = r["ub"][None]
    for mi in (0, int(r["iter"])):
        prm = sfb.QPSolverParams(max_iter=mi, polish=(which == 0))
        sfb.solve_qp_batch_host(P, q, A, l, u, prm)
        t0 = time.perf_counter(); sfb.solve_qp_batch_host(P, q, A, l, u, prm); dt = time.perf_counter() - t0
        print("   solve with max_iter %d: %.2f ms" % (mi, dt * 1e3))
