"""Per-iteration cost of the dense kernels: all QPs run the same number of iterations (max_iter fixed; stopping checks
off with stop_check_iter = 0, or on with the default 25), no polish; the difference between two caps isolates the ADMM
loop.  Prints microseconds per iteration of one wave (batch = one QP) and QP-iterations/s of a full batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
dev = torch.device("cuda:0")
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "20x40,32x32,32x64,40x60,64x64").split(",")]
for n, m in SIZES:
    for B in (1, int(os.environ.get("B", 4096))):
        P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
        d = [torch.from_numpy(a).to(dev) for a in (P, q, A, l, u)]
        x = torch.empty((B, n), dtype=torch.float64, device=dev); y = torch.empty((B, m), dtype=torch.float64, device=dev)
        obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream()
        def timed(prm):
            def go():
                sfb.solve_qp_batch_device(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), prm, stream=s.cuda_stream)
            go(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); go(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            return min(ts)
        res = []
        for sci in (0, 25):
            t0 = timed(sfb.QPSolverParams(max_iter=0, polish=False, stop_check_iter=sci, eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30))
            t1 = timed(sfb.QPSolverParams(max_iter=200, polish=False, stop_check_iter=sci, eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30))
            t2 = timed(sfb.QPSolverParams(max_iter=600, polish=False, stop_check_iter=sci, eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30))
            res.append((t0, (t2 - t1) / 400.0))
        print("(%d,%d) B %5d: setup %.3f ms | per iteration: no checks %.3f us, checks every 25: %.3f us -> %.3g QP-iter/s (no checks)" % (
            n, m, B, res[0][0], res[0][1] * 1e3, res[1][1] * 1e3, B / (res[0][1] * 1e-3)), flush=True)
