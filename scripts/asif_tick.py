"""ASIFSwarm ticks (host sensitivity ODE + assembly, one batched dense-QP solve on the GPU): wall time per tick."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from examples import models_lib as M
B = int(os.environ.get("B", 65536)); K = int(os.environ.get("K", 10))
M.asif_swarm_step(256, K, ticks=1)
for ticks in (1, 3):
    t0 = time.perf_counter(); out = M.asif_swarm_step(B, K, ticks=ticks); dt = time.perf_counter() - t0
    print("B=%d K=%d (n=3, m=%d): %d tick(s) in %.3f s; last tick iters mean %.1f max %d, codes %s" % (
        B, K, K + 3, ticks, dt, out["iter"].mean(), out["iter"].max(), np.bincount(out["code"], minlength=7)))
# the solve alone, on the QPs of the last tick
import smooth_feedback_amd as sfb
out = M.asif_swarm_step(B, K, ticks=1)
P, q, A, l, u = out["P"], out["q"], out["A"], out["l"], out["ub"]
prm = sfb.QPSolverParams(polish=False)   # examples/mpc_asif_vehicle.cpp:129 sets polish = false
for rep in range(3):
    t0 = time.perf_counter(); r = sfb.solve_qp_batch_host(P, q, A, l, u, prm); dt = time.perf_counter() - t0
    print("solve_qp_batch_host of the %d QPs: %.3f s (iters mean %.1f max %d)" % (B, dt, r.iter.mean(), r.iter.max()))
