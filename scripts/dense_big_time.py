"""Timing of the pivoted dense kernel for 64 < n+m <= 1024 (qp_dense_big.hip) on the sizes of the reference's ASIF
example / test and on square random problems: one QP (latency, host entry point) and a batch (device-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
dev = torch.device("cuda:0")
for n, m, B in ((3, 203, 4096), (4, 301, 2048), (40, 60, 4096), (100, 156, 1024), (200, 312, 256)):
    P, q, A, l, u = sfb.random_qp_batch(11, B, m, n, 0.6)
    u = u + 5.0
    prm = sfb.QPSolverParams(max_iter=2000)
    sfb.solve_qp_batch_host(P[:1], q[:1], A[:1], l[:1], u[:1], prm)
    t0 = time.perf_counter(); r1 = sfb.solve_qp_batch_host(P[:1], q[:1], A[:1], l[:1], u[:1], prm); t1 = time.perf_counter() - t0
    d = [torch.from_numpy(a).to(dev) for a in (P, q, A, l, u)]
    x = torch.empty((B, n), dtype=torch.float64, device=dev); y = torch.empty((B, m), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream()
    def go():
        sfb.solve_qp_batch_device(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), prm, stream=s.cuda_stream)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); go(); e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("n=%d m=%d: one QP %.2f ms (iter %d) | batch %d: %.1f ms -> %.0f QP/s, iters mean %.0f max %d, codes %s" % (
        n, m, t1 * 1e3, r1.iter[0], B, ms, B / ms * 1e3, it.float().mean().item(), it.max().item(), np.bincount(code.cpu().numpy(), minlength=7)))
