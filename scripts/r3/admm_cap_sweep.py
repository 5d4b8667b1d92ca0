"""Headline batch through the capped launch (SFB_SP_ADMM_CAP = waves inside the ADMM loop at a time; 0 = off)."""
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
prm = sfb.QPSolverParams()
def run():
    plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                            obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream)
def timed(reps=3):
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
os.environ["SFB_SP_ADMM_CAP"] = "0"
t = timed(); ref = (x.clone(), y.clone(), it.clone(), code.clone(), obj.clone())
print("no cap: %.2f ms -> %.0f QP/s" % (t, B / t * 1e3), flush=True)
for cap in os.environ.get("CAPS", "256,384,512,640,768,896,1024,1536").split(","):
    os.environ["SFB_SP_ADMM_CAP"] = cap
    t = timed()
    same = all(torch.equal(a, b) for a, b in zip(ref, (x, y, it, code, obj)))
    print("cap %5s: %.2f ms -> %.0f QP/s  identical %s" % (cap, t, B / t * 1e3, same), flush=True)
