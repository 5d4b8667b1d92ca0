import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import smooth_feedback_amd as sfb
B, m, n = 65536, 20, 10
P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
dev = torch.device("cuda:0")
t = [torch.from_numpy(a).to(dev) for a in (P, q, A, l, u)]
x = torch.empty((B, n), dtype=torch.float64, device=dev); y = torch.empty((B, m), dtype=torch.float64, device=dev)
obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
def run(prm):
    s = torch.cuda.current_stream()
    def go():
        sfb.solve_qp_batch_device(B, n, m, *[a.data_ptr() for a in t], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), prm, stream=s.cuda_stream)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); go(); e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
kw = dict(max_iter=1200, scaling=False, polish=False, eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30)
a = run(sfb.QPSolverParams(stop_check_iter=1, **kw))
for sci in (25, 50, 100, 300):
    b = run(sfb.QPSolverParams(stop_check_iter=sci, **kw))
    print("1200 iterations: no checks %.2f ms, checks every %d: %.2f ms -> +%.1f %% = %.2f us per check and wave" % (
        a, sci, b, 100 * (b - a) / a, (b - a) * 1e3 / (1200 / sci) / (B / 8192.0)))
