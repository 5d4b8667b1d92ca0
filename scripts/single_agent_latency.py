"""Latency of ONE MPC solve (the reference's own use: one controller, MPC::operator()) through the host entry
point: cold start and warm start, wall time and kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant = int(os.environ.get("VARIANT", 12)); K = int(os.environ.get("K", 50)); B = int(os.environ.get("B", 1))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
prm = sfb.QPSolverParams(max_iter=4000)
Px = np.tile(Pv, (B, 1)); q = np.zeros((B, d["n"]))
Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3)
Av2, l2, u2 = M.mpc_assemble_batch(variant, K, B, seed=4)
# like the MPC front: the explicit zeros of the transcription are declared (probed on a sample) unless NO_PRUNE=1
keep = None if os.environ.get("NO_PRUNE") == "1" else np.any(M.mpc_assemble_batch(variant, K, 64, seed=5)[0] != 0.0, axis=0)
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
r = plan.solve_batch_host(Px, q, Av, l, u, prm)
def wall(fn, reps=20):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * np.median(ts)
cold = wall(lambda: plan.solve_batch_host(Px, q, Av, l, u, prm))
warm = wall(lambda: plan.solve_batch_host(Px, q, Av2, l2, u2, prm, warm_x=r.primal, warm_y=r.dual))
r2 = plan.solve_batch_host(Px, q, Av2, l2, u2, prm, warm_x=r.primal, warm_y=r.dual)
nofac = wall(lambda: plan.solve_batch_host(Px, q, Av, l, u, sfb.QPSolverParams(max_iter=0, polish=False)))
print("B=%d variant %d K %d: cold %.2f ms (iters %s) | warm %.2f ms (iters %s) | scaling+factorisation only %.2f ms" % (
    B, variant, K, cold, r.iter[:4], warm, r2.iter[:4], nofac))
