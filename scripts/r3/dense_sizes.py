"""Dense QP throughput over the size class of BASELINE's north star (n <= 64): random QPs of benchmarks/bench_types.hpp
at (n, m) = (10,20) ... (64,64), two parameter sets (the reference benchmark's: eps 1e-6, no scaling, max_iter 10 000;
the library defaults with max_iter 10 000), device-resident, HIP-event timed; QP-iterations per second next to QP/s
(iteration counts differ a lot between sizes and parameter sets), parity of a sample against the dense oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from oracle import loader as O
dev = torch.device("cuda:0")
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "10x20,16x32,20x40,32x32,32x64,40x60,64x64").split(",")]
B0 = int(os.environ.get("B", 16384))
out = []
for n, m in SIZES:
    B = B0 if n + m <= 64 else max(2048, B0 // 4)
    P, q, A, l, u = sfb.random_qp_batch(5, B, m, n, 1.0)
    d = [torch.from_numpy(a).to(dev) for a in (P, q, A, l, u)]
    x = torch.empty((B, n), dtype=torch.float64, device=dev); y = torch.empty((B, m), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream()
    for name, prm, okw in (("bench", sfb.QPSolverParams(eps_abs=1e-6, eps_rel=1e-6, polish=True, max_iter=10000, scaling=False),
                            dict(eps_abs=1e-6, eps_rel=1e-6, polish=1, max_iter=10000, scaling=0)),
                           ("default", sfb.QPSolverParams(max_iter=10000), dict(max_iter=10000))):
        def go():
            sfb.solve_qp_batch_device(B, n, m, *[a.data_ptr() for a in d], x.data_ptr(), y.data_ptr(), obj.data_ptr(), it.data_ptr(), code.data_ptr(), prm, stream=s.cuda_stream)
        go(); torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); go(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        itc = it.cpu().numpy().astype(np.int64); cd = code.cpu().numpy()
        S = min(B, 256 if n + m > 64 else 512)
        ref = O.qp_dense_solve_batch(P[:S], q[:S], A[:S], l[:S], u[:S], params=O.default_params(**okw), nthreads=min(64, os.cpu_count() or 1))
        fin = np.isfinite(ref["x"]).all(axis=1)
        par = dict(sample=S, code_mismatches=int((cd[:S] != ref["code"]).sum()), iter_mismatches=int((itc[:S] != ref["iter"]).sum()),
                   max_abs_dx=float(np.abs(x[:S].cpu().numpy() - ref["x"])[fin].max(initial=0.0)))
        rec = dict(n=n, m=m, params=name, batch=B, ms=ms, qp_per_s=B / ms * 1e3, qp_iterations_per_s=float(itc.sum()) / ms * 1e3,
                   iter_mean=float(itc.mean()), iter_max=int(itc.max()), codes=np.bincount(cd, minlength=7).tolist(), parity=par)
        out.append(rec)
        print("(%2d,%2d) %-7s B %5d: %8.2f ms  %9.0f QP/s  %.3g QP-iter/s  iters mean %.0f max %d codes %s parity %s" % (
            n, m, name, B, ms, rec["qp_per_s"], rec["qp_iterations_per_s"], rec["iter_mean"], rec["iter_max"], rec["codes"], par), flush=True)
if os.environ.get("OUT"):
    json.dump(out, open(os.environ["OUT"], "w"), indent=1)
