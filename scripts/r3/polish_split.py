"""Polish of the headline model (pruned plan): a lone wave and the 8 192-agent batch, with polish_iter = 5 (default) and 0
(second factorisation only), against no polish at all."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K = 12, 50
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for B in (1, 8192):
    Av, l, u = M.mpc_assemble_batch(variant, K, max(B, 64), seed=3, threads=64)
    keep = np.any(Av[:: max(1, len(Av) // 64)] != 0.0, axis=0)
    Av, l, u = Av[:B], l[:B], u[:B]
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
    x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream()
    def timed(prm):
        def go():
            plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                                    obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream)
        go(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); go(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)
    a = timed(sfb.QPSolverParams(polish=False)); b = timed(sfb.QPSolverParams(polish=True, polish_iter=0)); c = timed(sfb.QPSolverParams())
    print("B %5d: no polish %.3f ms | + second factorisation %.3f ms | + 5 refinement steps %.3f ms  (whole solve %.3f ms)" % (B, a, b - a, c - b, c), flush=True)
