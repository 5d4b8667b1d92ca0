"""Workload of the rocprofv3 passes for the on-chip dense kernel of 32 < n + m <= 128 (qp_dense_mid.hip): every size of the
bench line's size table at a batch larger than the chip holds (the split launch: setup / time-sliced loop / finish kernels),
reference-benchmark parameters, max_iter 2000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
dev = torch.device("cuda:0")
for n, m, B in ((16, 32, 16384), (20, 40, 16384), (32, 64, 8192), (40, 60, 8192), (64, 64, 8192)):
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
    d = [torch.from_numpy(a).to(dev) for a in (P, q, A, l, u)]
    x = torch.empty((B, n), dtype=torch.float64, device=dev); y = torch.empty((B, m), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
    prm = sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=2000, scaling=False)
    for _ in range(2):
        sfb.solve_qp_batch_device(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), prm)
    torch.cuda.synchronize()
    print("(%d,%d) B %d: iterations mean %.0f" % (n, m, B, it.float().mean().item()))
