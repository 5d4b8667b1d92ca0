"""Where a lone dense QP spends its time outside the ADMM loop: setup without / with scaling (max_iter = 0), polish
(converged solve with polish on minus off), for one wave and for a batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
dev = torch.device("cuda:0")
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "16x32,20x40,32x32,32x64,40x60,64x64").split(",")]
for n, m in SIZES:
    for B in (1, int(os.environ.get("B", 8192))):
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
        s0 = timed(sfb.QPSolverParams(max_iter=0, polish=False, scaling=False))
        s1 = timed(sfb.QPSolverParams(max_iter=0, polish=False, scaling=True))
        # polish: library tolerances, enough iterations for most QPs to converge; same iterations with and without
        p0 = timed(sfb.QPSolverParams(max_iter=2000, polish=False, scaling=False))
        itn = it.cpu().numpy().astype(np.int64); cd = code.cpu().numpy()
        p1 = timed(sfb.QPSolverParams(max_iter=2000, polish=True, scaling=False))
        print("(%d,%d) B %5d: setup %.3f ms, with scaling %.3f ms | solve %.3f ms, with polish %.3f ms (optimal %d of %d, mean iterations %.0f)" % (
            n, m, B, s0, s1, p0, p1, int((cd == 0).sum()), B, itn.mean()), flush=True)
