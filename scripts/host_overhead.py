"""Host-pointer entry point vs device-resident entry point of the sparse path (8192 MPC agents): what malloc + PCIe cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K, B = 12, 50, int(os.environ.get("B", 8192))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
t0 = time.perf_counter(); Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64); print("assembly %.3f s" % (time.perf_counter() - t0))
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K))
Px = np.tile(Pv, (B, 1)); q = np.zeros((B, d["n"]))
prm = sfb.QPSolverParams(max_iter=40, stop_check_iter=1000, polish=True)   # a warm-tick-like amount of work
for rep in range(3):
    t0 = time.perf_counter(); r = plan.solve_batch_host(Px, q, Av, l, u, prm); th = time.perf_counter() - t0
    print("host entry %.3f s" % th)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t0 = time.perf_counter(); dPx, dq, dAx, dl, du = T(Px), T(q), T(Av), T(l), T(u); torch.cuda.synchronize(); print("torch H2D (pageable) %.3f s for %.2f GB" % (time.perf_counter() - t0, (Px.nbytes + q.nbytes + Av.nbytes + l.nbytes + u.nbytes) / 1e9))
pin = [torch.from_numpy(a).pin_memory() for a in (Px, q, Av, l, u)]
t0 = time.perf_counter(); [a.to(dev, non_blocking=True) for a in pin]; torch.cuda.synchronize(); print("torch H2D (pinned) %.3f s" % (time.perf_counter() - t0))
x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
t0 = time.perf_counter(); ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev); torch.cuda.synchronize(); print("workspace alloc %.3f s (%.1f GB)" % (time.perf_counter() - t0, ws.numel() * 8 / 1e9))
s = torch.cuda.current_stream()
for rep in range(2):
    t0 = time.perf_counter()
    plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream)
    torch.cuda.synchronize(); print("device entry %.3f s" % (time.perf_counter() - t0))
