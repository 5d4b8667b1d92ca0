"""Timeline of one launch of the sparse MPC solve from the per-item wall-clock stamps of the profiling build
(PROF_DEFS=-DSFB_SP_TIMELINE scripts/build_prof.sh; run with SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so):
resident items over time, time per phase, iteration rate of the long runners."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K, B = 12, 50, int(os.environ.get("B", 8192))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
keep = None if os.environ.get("NO_PRUNE") == "1" else np.any(Av[:: max(1, B // 64)] != 0.0, axis=0)
plan = sfb.SparseQPPlan(d["n"], d["m"], Pp, Pi, Ap, Aj, stage=M.mpc_stage(variant, K), keep=keep)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dPx, dq, dAx, dl, du = T(np.tile(Pv, (B, 1))), T(np.zeros((B, d["n"]))), T(Av), T(l), T(u)
x = torch.empty((B, d["n"]), dtype=torch.float64, device=dev); y = torch.empty((B, d["m"]), dtype=torch.float64, device=dev)
obj = torch.empty(B, dtype=torch.float64, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev); code = torch.empty(B, dtype=torch.int32, device=dev)
ws = torch.empty((plan.workspace_bytes(B) + 7) // 8, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream()
for _ in range(2):
    plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                            obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), sfb.QPSolverParams(), stream=s.cuda_stream)
    torch.cuda.synchronize()
st = y[:, :4].cpu().numpy() / 100.0  # microseconds (100 MHz)
itc = it.cpu().numpy()
t0 = st[:, 0].min()
st -= t0
end = st[:, 3].max()
print("launch %.2f ms; items %d; iterations mean %.1f max %d" % (end / 1e3, B, itc.mean(), itc.max()))
print("phase means (us): setup+factor %.0f | ADMM loop %.0f | polish+report %.0f" % ((st[:, 1] - st[:, 0]).mean(), (st[:, 2] - st[:, 1]).mean(), (st[:, 3] - st[:, 2]).mean()))
grid = np.linspace(0, end, 41)
print("t(ms) resident  in-setup in-loop in-polish")
for g in grid:
    res = (st[:, 0] <= g) & (st[:, 3] > g)
    a = res & (st[:, 1] > g); b = res & (st[:, 1] <= g) & (st[:, 2] > g); c = res & (st[:, 2] <= g)
    print("%6.1f %8d %8d %8d %8d" % (g / 1e3, res.sum(), a.sum(), b.sum(), c.sum()))
long = np.argsort(-itc)[:8]
for b in long:
    print("item %5d: iter %4d start %.1f ms loop %.1f ms -> %.1f us/iteration, polish %.2f ms, end %.1f ms" % (
        b, itc[b], st[b, 0] / 1e3, (st[b, 2] - st[b, 1]) / 1e3, (st[b, 2] - st[b, 1]) / max(1, itc[b]), (st[b, 3] - st[b, 2]) / 1e3, st[b, 3] / 1e3))
# per-iteration speed of an item as a function of when it ran: items grouped by start time
q = np.quantile(st[:, 0], [0, .25, .5, .75, 1.0])
for lo, hi in zip(q[:-1], q[1:]):
    sel = (st[:, 0] >= lo) & (st[:, 0] <= hi) & (itc > 0)
    print("start in [%.1f, %.1f] ms: %5d items, mean us/iteration %.1f" % (lo / 1e3, hi / 1e3, sel.sum(), ((st[sel, 2] - st[sel, 1]) / itc[sel]).mean()))
