import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
B = int(os.environ.get("B", 8192))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(12, 50)
Av, l, u = M.mpc_assemble_batch(12, 50, B, seed=3, threads=64)
keep = None if os.environ.get("NO_PRUNE") == "1" else np.any(Av[:: max(1, B // 64)] != 0.0, axis=0)
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(12, 50), keep=keep)
r = plan.solve_batch_host(np.tile(Pv, (B, 1)), np.zeros((B, d["n"])), Av, l, u, sfb.QPSolverParams(max_iter=0, polish=False))
print("done", B)
