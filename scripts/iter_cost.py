"""Per-iteration cost of the dense QP kernel: fixed iteration counts (stop_check_iter=1 never checks,
qp_solver.hpp:465), two max_iter values, difference -> SIMD-cycles per ADMM iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb

B, m, n = 65536, int(os.environ.get("M", 20)), int(os.environ.get("N", 10))
P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
dev = torch.device("cuda:0")
t = [torch.from_numpy(a).to(dev) for a in (P, q, A, l, u)]
x = torch.empty((B, n), dtype=torch.float64, device=dev); y = torch.empty((B, m), dtype=torch.float64, device=dev)
obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
def run(maxit, sci=1):
    prm = sfb.QPSolverParams(max_iter=maxit, stop_check_iter=sci, scaling=False, polish=False)
    s = torch.cuda.current_stream()
    def go():
        sfb.solve_qp_batch_device(B, n, m, *[a.data_ptr() for a in t], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), prm, stream=s.cuda_stream)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); go(); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
t0 = run(0); t1 = run(200); t2 = run(1200)
per_iter_ns = (t2 - t1) * 1e6 / 1000 / B   # ns per QP-iteration, whole chip
print("setup-only %.3f ms | 200 it %.3f ms | 1200 it %.3f ms | %.4f ns/iter/QP chip-wide -> %.0f SIMD-cycles/iter @2.4GHz x1024 SIMDs"
      % (t0, t1, t2, per_iter_ns, per_iter_ns * 2.4 * 1024))
t3 = run(1200, sci=25)
print("with checks every 25 (may stop early): %.3f ms" % t3)
