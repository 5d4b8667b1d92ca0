"""Per-iteration time of the LAT form of the sparse ADMM loop (whole launch in the LAT form, no stopping checks, fixed iteration
counts) for a lone wave and for one / two / three LAT waves per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K = 12, 50
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
for k, v in (("SFB_SP_PREDICT", "0"), ("SFB_SP_FORCE_LAT", "1"), ("SFB_SP_LEAN_WAVES", "1000000000")):
    sfb.debug_set(k, v)
AvA, lA, uA = M.mpc_assemble_batch(variant, K, 768, seed=3, threads=64)
keep = np.any(AvA != 0.0, axis=0)
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for B in (1, 256, 512, 768):
    Av, l, u = AvA[:B], lA[:B], uA[:B]
    dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
    x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
    obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream()
    def timed(prm):
        def go():
            plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                                    obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream)
        go(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); go(); e1.record(s); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)
    tiny = dict(eps_abs=1e-30, eps_rel=1e-30, eps_primal_inf=1e-30, eps_dual_inf=1e-30, polish=False)
    a = timed(sfb.QPSolverParams(max_iter=101, stop_check_iter=0, **tiny)); b = timed(sfb.QPSolverParams(max_iter=401, stop_check_iter=0, **tiny))
    per = (b - a) / 300 * 1e3
    print("B %5d: %.2f us per iteration of every item -> %.1f item-iterations/us" % (B, per, B / per), flush=True)
