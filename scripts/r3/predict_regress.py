"""Launch in predicted order against the single time-sliced kernel (SFB_SP_PREDICT=0): batch sizes, cold and warm start,
two problem sizes.  Results are compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for variant, K, sizes in ((12, 50, (3072, 4096, 8192, 16384, 32768)), (6, 30, (8192, 32768)), (6, 10, (16384, 65536))):
    d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
    for B in sizes:
        Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
        keep = np.any(Av[:: max(1, B // 64)] != 0.0, axis=0)
        plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
        dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
        mk = lambda: (torch.empty((B, d["n"]), dtype=torch.float64, device=dev), torch.empty((B, d["m"]), dtype=torch.float64, device=dev),
                      torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev))
        ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
        s = torch.cuda.current_stream()
        prm = sfb.QPSolverParams()
        def run(out, warm=None):
            x, y, obj, it, code = out
            plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                                    obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream,
                                    dwarm_x=warm[0].data_ptr() if warm else 0, dwarm_y=warm[1].data_ptr() if warm else 0)
        def timed(out, warm=None, reps=3):
            run(out, warm); torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); run(out, warm); e1.record(s); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        res = {}
        for mode in ("0", "1"):
            os.environ["SFB_SP_PREDICT"] = mode
            cold = mk(); tc = timed(cold)
            # warm start as a swarm tick sees it: the previous solution, bounds moved a little
            warm_out = mk(); tw = timed(warm_out, warm=(cold[0], cold[1]))
            res[mode] = (tc, tw, cold, warm_out)
        a, b = res["0"], res["1"]
        same = all(torch.equal(p, q) for p, q in zip(a[2], b[2])) and all(torch.equal(p, q) for p, q in zip(a[3], b[3]))
        print("variant %2d K %2d (n %4d) B %6d | cold: single %.2f ms, predicted %.2f ms (%+.1f %%) | warm (mean iter %.1f): single %.2f, predicted %.2f (%+.1f %%) | identical %s" % (
            variant, K, d["n"], B, a[0], b[0], 100 * (b[0] / a[0] - 1), b[3][3].float().mean().item(), a[1], b[1], 100 * (b[1] / a[1] - 1), same), flush=True)
        del ws
