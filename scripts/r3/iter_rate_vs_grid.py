"""Steady-state ADMM iteration rate of the sparse kernel as a function of the number of resident waves
(SFB_SP_GRID) and of the load flavour of the factor stream (SFB_SP_LEAN_WAVES): all items run the same number
of iterations (max_iter fixed, no stopping checks, no polish), so there is no drain; the difference between two
iteration caps isolates the loop.  Prints item-iterations per microsecond."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant = int(os.environ.get("VARIANT", 12)); K = int(os.environ.get("K", 50)); B = int(os.environ.get("B", 8192))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
keep = np.any(Av[:: max(1, B // 64)] != 0.0, axis=0)
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
def timed(prm, reps=2):
    run(prm); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(prm); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
lo, hi = int(os.environ.get("IT_LO", 40)), int(os.environ.get("IT_HI", 140))
os.environ["SFB_SP_SLICE"] = "1000000"
for wpc in os.environ.get("WPC", "0").split(","):
    if wpc != "0": os.environ["SFB_SP_WAVES_PER_CU"] = wpc
    else: os.environ.pop("SFB_SP_WAVES_PER_CU", None)
    for lean in os.environ.get("LEAN", "-1,100000000").split(","):
        os.environ["SFB_SP_LEAN_WAVES"] = lean
        for g in [int(g) for g in os.environ.get("GRIDS", "256,512,768,1024,1536,2048,3072").split(",")]:
            os.environ["SFB_SP_GRID"] = str(g)
            t0 = timed(sfb.QPSolverParams(max_iter=0, polish=False))
            t1 = timed(sfb.QPSolverParams(max_iter=lo, stop_check_iter=1, polish=False))
            t2 = timed(sfb.QPSolverParams(max_iter=hi, stop_check_iter=1, polish=False))
            rate = B * (hi - lo) / ((t2 - t1) * 1e3)
            print("wpc %s lean %s grid %5d: setup %.2f ms | %d it %.2f ms | %d it %.2f ms | %.2f item-iter/us, %.1f us per iteration of a wave" % (
                wpc, lean, g, t0, lo, t1, hi, t2, rate, min(g, B) / rate), flush=True)
