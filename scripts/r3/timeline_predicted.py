"""Timeline of a launch in predicted order from the per-item stamps of the profiling build
(PROF_DEFS=-DSFB_SP_TIMELINE scripts/build_prof.sh; SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so): when the second
launch takes each survivor up, how fast the long ones iterate there, how many items are active over time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import smooth_feedback_amd as sfb
from examples import models_lib as M
variant, K, B = 12, 50, int(os.environ.get("B", 8192))
d, Pp, Pi, Pv, Ap, Aj = M.mpc_pattern(variant, K)
Av, l, u = M.mpc_assemble_batch(variant, K, B, seed=3, threads=64)
keep = np.any(Av[:: max(1, B // 64)] != 0.0, axis=0)
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
st = y[:, :4].cpu().numpy() / 100.0  # us
itc = it.cpu().numpy()
st -= st[:, 0].min()
surv = itc > 27
k1_end = st[~surv, 3].max()
print("launch %.2f ms; first launch ends ~%.2f ms (last item finished there); survivors %d" % (st[:, 3].max() / 1e3, k1_end / 1e3, surv.sum()))
order = np.argsort(-itc)
print("rank  item  iter | taken up (ms) | loop in 2nd launch (ms) | us/iteration there | polish+report (ms) | end (ms)")
for r in list(range(0, 12)) + [16, 32, 64, 128, 256, 400, 512, 640, 700, 800, 1000, 1500, 2000, 3000, 4000]:
    if r >= surv.sum(): break
    b = order[r]
    print("%4d %5d %5d | %8.2f | %8.2f | %8.1f | %6.2f | %8.2f" % (r, b, itc[b], st[b, 1] / 1e3, (st[b, 2] - st[b, 1]) / 1e3, (st[b, 2] - st[b, 1]) / (itc[b] - 27),
                                                                     (st[b, 3] - st[b, 2]) / 1e3, st[b, 3] / 1e3))
t_a = st[surv, 1].min()
print("second launch: first item taken up at %.2f ms" % (t_a / 1e3))
print("t(ms)  in-loop  in-polish   (survivors only)")
for g in np.arange(t_a, st[:, 3].max() + 1, 2000.0):
    a = surv & (st[:, 1] <= g) & (st[:, 2] > g); c = surv & (st[:, 2] <= g) & (st[:, 3] > g)
    print("%6.1f %8d %8d" % (g / 1e3, a.sum(), c.sum()))
