"""Cost of one stopping check of the sparse kernel for a lone wave and for the full batch: fixed iteration count,
tolerances that never fire, stop_check_iter 25 vs never."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K = 12, 50
for B in (1, 8192):
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
    plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
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
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); go(); e1.record(s); torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    N = 500
    kw = dict(max_iter=N, polish=False, eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30)
    t_no = timed(sfb.QPSolverParams(stop_check_iter=1, **kw))
    t_ck = timed(sfb.QPSolverParams(stop_check_iter=25, **kw))
    print("B=%d: %d iterations without checks %.2f ms, with a check every 25: %.2f ms -> %.3f ms per check (= %.1f iterations)"
          % (B, N, t_no, t_ck, (t_ck - t_no) / (N // 25), (t_ck - t_no) / (N // 25) / ((t_no) / N)))
