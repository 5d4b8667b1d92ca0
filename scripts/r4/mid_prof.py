"""Phase times of ONE dense QP in the on-chip kernel (profiling build: scripts/r4/build_variant.sh prof qp_dense_mid.hip -DSFB_MID_PROF;
run with SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import smooth_feedback_amd as sfb
for s in os.environ.get("SIZES", "16x32,20x40,32x64,40x60,64x64").split(","):
    n, m = (int(v) for v in s.split("x"))
    B = int(os.environ.get("B", 1))
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
    for scaling in (False, True):
        r = sfb.solve_qp_batch_host(P, q, A, l, u, sfb.QPSolverParams(max_iter=2000, scaling=scaling))
        sys.stdout.flush()
