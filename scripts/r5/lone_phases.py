"""Phase times of ONE MPC QP solved alone on the chip (lone wave) and inside the full batch, for several polish_iter:
separates the polish factorisation from its refinement rounds.  sfb_sparse_qp_solve_batch_phases (TRACE instance)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K = 12, 50
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
Av0, _, _ = M.mpc_assemble_batch(variant, K, 64, seed=3, threads=64)
if os.environ.get("PLAN_DEBUG") == "1": sfb.debug_set("SFB_PLAN_DEBUG", "1")
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=np.any(Av0 != 0.0, axis=0))  # (pruned like bench.py)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
names = ["scale", "fill", "factor", "iterate", "polish", "report"]
for B in (1, 64, 8192):
    Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
    dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
    x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
    ph = torch.zeros((B, 6), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream()
    for pi in (5, 1, 0):
        prm = sfb.QPSolverParams(max_iter=4000, polish_iter=pi)
        for _ in range(2):
            plan.solve_batch_device_phases(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                                           obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), ph.data_ptr(), prm, stream=s.cuda_stream)
            torch.cuda.synchronize()
        p = ph.cpu().numpy(); iters = it.cpu().numpy()
        print("B=%5d polish_iter=%d  mean iterations %.1f  us per item: %s   iterate/iteration %.2f us" % (
            B, pi, iters.mean(), "  ".join("%s %.0f" % (n, v) for n, v in zip(names, p.mean(0))), p[:, 3].sum() / iters.sum()))
