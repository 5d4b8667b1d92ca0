"""Throughput of the MPC batch solve when consecutive, independent batches are launched on S streams (each with its
own workspace and outputs): the tail of one launch (only long-running agents left) overlaps with the bulk of the next."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K, B, STEPS = 12, 50, 8192, int(os.environ.get("STEPS", 8))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
prm = sfb.QPSolverParams(max_iter=4000)
def bufs():
    return dict(x=torch.empty((B, d["n"]), dtype=torch.float64, device=dev), y=torch.empty((B, d["m"]), dtype=torch.float64, device=dev),
                it=torch.empty(B, dtype=torch.int32, device=dev), code=torch.empty(B, dtype=torch.int32, device=dev),
                ws=torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev))
for S in (1, 2, 3):
    sets = [bufs() for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def launch(i):
        b, s = sets[i % S], streams[i % S]
        plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), b["x"].data_ptr(),
                                b["y"].data_ptr(), 0, b["it"].data_ptr(), b["code"].data_ptr(), b["ws"].data_ptr(), prm, stream=s.cuda_stream)
    for i in range(S): launch(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS): launch(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all((b["code"] == 0).all().item() for b in sets)
    print("streams=%d: %d batches in %.1f ms -> %.1f ms per batch, %.0f solves/s (all Optimal: %s)" % (S, STEPS, dt * 1e3, dt * 1e3 / STEPS, B * STEPS / dt, ok))
    del sets
