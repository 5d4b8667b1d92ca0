"""Dense QP batch with the LIBRARY DEFAULT parameters (eps 1e-3, scaling, polish): short ADMM loops, so problem
setup (scaling, pivoted LDL') weighs much more than in the benchmark-parameter batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
B, m, n = 65536, int(os.environ.get("M", 20)), int(os.environ.get("N", 10))
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
CAP = int(os.environ.get("CAP", 500))   # unbounded max_iter lets a handful of random problems run for millions of iterations
for name, prm in (("default, max_iter=%d" % CAP, sfb.QPSolverParams(max_iter=CAP)), ("same, no polish", sfb.QPSolverParams(max_iter=CAP, polish=False)),
                  ("max_iter=50", sfb.QPSolverParams(max_iter=50)), ("setup only", sfb.QPSolverParams(max_iter=0))):
    ms = run(prm)
    print("%-20s %.3f ms -> %.2f M QP/s ; iters mean %.1f ; codes %s" % (name, ms, B / ms / 1e3, it.float().mean().item(), np.bincount(code.cpu().numpy(), minlength=7)))
