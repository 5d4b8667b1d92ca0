"""Phase times of the sparse MPC solve (median of REPS launches each): setup+factor, 100 iterations without checks,
full solve without polish, full solve.  Use with SFB_LIB_PATH to compare builds on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K, B, REPS = 12, 50, int(os.environ.get("B", 8192)), int(os.environ.get("REPS", 5))
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
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); go(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
P = sfb.QPSolverParams
setup = timed(P(max_iter=0, polish=False))
it100 = timed(P(max_iter=100, stop_check_iter=1000000, polish=False))
chk100 = timed(P(max_iter=100, stop_check_iter=25, eps_abs=1e-30, eps_rel=1e-30, polish=False))
nopol = timed(P(max_iter=4000, polish=False))
full = timed(P(max_iter=4000))
print("%s: setup+factor %.2f | 100 iterations %.2f (+4 checks %.2f) | solve without polish %.2f | full %.2f ms" % (
    os.path.basename(os.environ.get("SFB_LIB_PATH", "libsfb.so")), setup, it100 - setup, chk100 - it100, nopol, full))
