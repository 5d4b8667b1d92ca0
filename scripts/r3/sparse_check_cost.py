"""Cost of a stopping check of the sparse kernel for a lone wave and in a full batch: fixed iteration counts, tolerances
that never trigger, checks every 25 iterations vs none."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K = 12, 50
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
for B in (1, 8192):
    Av, l, u = M.mpc_assemble_batch(variant, K, max(B, 64), seed=3, threads=64)
    keep = np.any(Av != 0.0, axis=0)
    Av, l, u = Av[:B], l[:B], u[:B]
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
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
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); go(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)
    tiny = dict(eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30, polish=False)
    os.environ["SFB_SP_SLICE"] = "1000000"; os.environ["SFB_SP_PREDICT"] = "0"
    a = timed(sfb.QPSolverParams(max_iter=101, stop_check_iter=0, **tiny)); b = timed(sfb.QPSolverParams(max_iter=401, stop_check_iter=0, **tiny))
    c = timed(sfb.QPSolverParams(max_iter=101, stop_check_iter=25, **tiny)); e = timed(sfb.QPSolverParams(max_iter=401, stop_check_iter=25, **tiny))
    it0 = (b - a) / 300; it1 = (e - c) / 300
    print("B %5d: per iteration without checks %.2f us, with a check every 25: %.2f us -> one check = %.1f us = %.1f iterations" % (
        B, it0 * 1e3, it1 * 1e3, (it1 - it0) * 25 * 1e3, (it1 - it0) * 25 / it0))
