"""Timing of the sparse MPC QP kernel on device-resident data (variant 12, K=50 by default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant = int(os.environ.get("VARIANT", 12)); K = int(os.environ.get("K", 50)); B = int(os.environ.get("B", 2048))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
t0 = time.time(); Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64); ta = time.time() - t0
keep = None if os.environ.get("NO_PRUNE") == "1" else np.any(Av[:: max(1, B // 64)] != 0.0, axis=0)
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream()
def run(prm):
    plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                            obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream)
def timed(prm):
    run(prm); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); run(prm); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
print("variant", variant, "K", K, "B", B, "n=m", d["n"], "nnzL", plan.nnzL, "nnzA analysed", plan.nnzA_kept, "of", plan.nnzA, "host assembly s", round(ta, 3))
ms = timed(sfb.QPSolverParams(max_iter=4000))
print("default params: %.2f ms -> %.0f MPC-QP solves/s ; iters mean %.1f max %d ; codes %s" % (ms, B / ms * 1e3, it.float().mean().item(), it.max().item(), np.bincount(code.cpu().numpy(), minlength=7)))
t_setup = timed(sfb.QPSolverParams(max_iter=0, polish=False))
t50 = timed(sfb.QPSolverParams(max_iter=50, stop_check_iter=1, polish=False))
t100 = timed(sfb.QPSolverParams(max_iter=100, stop_check_iter=1, polish=False))
per_it = (t100 - t50) / 50
bytes_it = 2 * plan.nnzL * 8 * B
print("setup+factor %.2f ms | per ADMM iteration %.3f ms (whole batch) -> factor stream %.1f GB/s" % (t_setup, per_it, bytes_it / per_it / 1e6))
itc = it.cpu().numpy()
print("iteration percentiles 50/90/99/99.9/max:", [int(np.percentile(itc, p)) for p in (50, 90, 99, 99.9, 100)], " #>300:", int((itc > 300).sum()))
for cap in (200, 400):
    ms = timed(sfb.QPSolverParams(max_iter=cap))
    print("max_iter=%d: %.2f ms -> %.0f solves/s ; codes %s" % (cap, ms, B / ms * 1e3, np.bincount(code.cpu().numpy(), minlength=7)))
print("breakdown: setup(scaling on) %.2f | scaling off %.2f ms" % (timed(sfb.QPSolverParams(max_iter=0, polish=False)), timed(sfb.QPSolverParams(max_iter=0, polish=False, scaling=False))))
# factor reuse (sfb_qp_params::reuse_factor): the same matrices again, as a time-invariant MPC presents them every tick
ms_plain = timed(sfb.QPSolverParams(max_iter=4000))
ms_reuse = timed(sfb.QPSolverParams(max_iter=4000, reuse_factor=True))
print("same matrices again: %.2f ms without reuse_factor, %.2f ms with (%.0f solves/s)" % (ms_plain, ms_reuse, B / ms_reuse * 1e3))
print("   setup only (max_iter 0, no polish): %.2f ms without, %.2f ms with" % (timed(sfb.QPSolverParams(max_iter=0, polish=False)), timed(sfb.QPSolverParams(max_iter=0, polish=False, reuse_factor=True))))
