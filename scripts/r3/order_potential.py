"""How much of the headline launch is a SCHEDULING problem?  The batch is launched (a) in natural order, (b) longest
item first with the true iteration counts of a previous run as the (perfect) predictor, for several grids / slices /
phased variants.  Results never depend on the order (checked)."""
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
prm = sfb.QPSolverParams()
def run(order=None):
    plan.solve_batch_device(B, dPx.data_ptr(), dq.data_ptr(), dAx.data_ptr(), dl.data_ptr(), du.data_ptr(), x.data_ptr(), y.data_ptr(),
                            obj.data_ptr(), it.data_ptr(), code.data_ptr(), ws.data_ptr(), prm, stream=s.cuda_stream,
                            dorder=order.data_ptr() if order is not None else 0)
def timed(order=None, reps=3):
    run(order); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); run(order); e1.record(s); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
def env(**kw):
    for k in ("SFB_SP_SLICE", "SFB_SP_GRID", "SFB_SP_PHASED", "SFB_SP_GRID2", "SFB_SP_LEAN_WAVES", "SFB_SP_LEAN_WAVES2", "SFB_SP_PREDICT",
              "SFB_SP_GRID3", "SFB_SP_LEAN_WAVES3", "SFB_SP_SLICE3", "SFB_SP_PAUSE", "SFB_SP_CRIT", "SFB_SP_DEEP3", "SFB_SP_NAP"): os.environ.pop(k, None)
    for k, v in kw.items(): os.environ[k] = str(v)
env(SFB_SP_PREDICT=0)
t_nat = timed()
itc = it.cpu().numpy().copy(); x0 = x.clone(); c0 = code.clone()
print("natural order: %.2f ms ; iterations mean %.1f p50 %d p90 %d p99 %d max %d ; sum %d" % (t_nat, itc.mean(), *[int(np.percentile(itc, p)) for p in (50, 90, 99, 100)], itc.sum()))
for thr in (50, 100, 150, 200, 300, 400, 600):
    print("   items with more than %4d iterations: %5d, their work beyond it %8d item-iterations (%.1f %%)" % (thr, (itc > thr).sum(), np.maximum(itc - thr, 0).sum(), 100.0 * np.maximum(itc - thr, 0).sum() / itc.sum()))
longest = torch.from_numpy(np.argsort(-itc, kind="stable").astype(np.int32)).to(dev)
rng = np.random.default_rng(0)
noisy = {}
for sig in (0.3, 0.6, 1.0):  # a predictor with log-normal error of that sigma
    noisy[sig] = torch.from_numpy(np.argsort(-(itc * np.exp(sig * rng.standard_normal(B))), kind="stable").astype(np.int32)).to(dev)
def check():
    assert torch.equal(code, c0) and np.array_equal(it.cpu().numpy(), itc) and torch.equal(x, x0), "results depend on the order!"
if os.environ.get("PART", "predict") == "nap":
    for g in (640, 768, 1024):
        for crit in (64, 128, 256):
            for nap in (0, 4, 8, 16, 32):
                env(SFB_SP_GRID3=g, SFB_SP_CRIT=crit, SFB_SP_NAP=nap)
                a = timed(reps=2); check()
                print("grid3 %4d, first %3d items unpaced, the others nap %2d x 0.43 us per iteration: %.2f ms (%.0f QP/s)" % (g, crit, nap, a, B / a * 1e3), flush=True)
    sys.exit(0)
if os.environ.get("PART", "predict") == "deep":
    for deep in (1, 0):
        for g in (256, 320, 384, 448, 512, 576, 640, 768):
            env(SFB_SP_GRID3=g, SFB_SP_DEEP3=deep)
            a = timed(); check()
            print("grid3 %4d, prefetch distance %2d units: %.2f ms (%.0f QP/s)" % (g, 16 if deep else 8, a, B / a * 1e3), flush=True)
    sys.exit(0)
if os.environ.get("PART", "predict") == "crit":
    for g in (640, 768, 1024, 1536, 2560):
        for crit in (0, 64, 128, 256):
            env(SFB_SP_GRID3=g, SFB_SP_LEAN_WAVES3=0, SFB_SP_CRIT=crit)
            a = timed(); check()
            print("grid3 %4d, all but the first %3d items with non-temporal loads: %.2f ms (%.0f QP/s)" % (g, crit, a, B / a * 1e3), flush=True)
    sys.exit(0)
if os.environ.get("PART", "predict") == "predict":
    for kw in ([dict(SFB_SP_GRID3=g) for g in (320, 384, 448, 512, 576, 640, 768, 1024, 1536)] +
               [dict(SFB_SP_GRID3=g, SFB_SP_LEAN_WAVES3=0) for g in (512, 768)] +
               [dict(SFB_SP_GRID3=512, SFB_SP_PAUSE=p) for p in (2, 52, 77)] +
               [dict(SFB_SP_GRID3=512, SFB_SP_SLICE3=s) for s in (100, 200)] + [dict()]):
        env(**kw)
        a = timed(); check()
        print("predicted order %-60s %.2f ms (%.0f QP/s)" % (kw, a, B / a * 1e3), flush=True)
    sys.exit(0)
for name, kw in (("default (slice 50, full grid)", dict(SFB_SP_PREDICT=0)),
                 ("no slicing (FCFS), full grid", dict(SFB_SP_SLICE=100000)),
                 ("FCFS grid 2048", dict(SFB_SP_SLICE=100000, SFB_SP_GRID=2048)),
                 ("FCFS grid 1536", dict(SFB_SP_SLICE=100000, SFB_SP_GRID=1536)),
                 ("FCFS grid 1024", dict(SFB_SP_SLICE=100000, SFB_SP_GRID=1024)),
                 ("slice 50 grid 1536", dict(SFB_SP_GRID=1536)),
                 ("slice 200 full grid", dict(SFB_SP_SLICE=200)),
                 ("phased, FCFS, grid2 768", dict(SFB_SP_PHASED=1, SFB_SP_SLICE=100000, SFB_SP_GRID2=768)),
                 ("phased, FCFS, grid2 1024", dict(SFB_SP_PHASED=1, SFB_SP_SLICE=100000, SFB_SP_GRID2=1024)),
                 ("phased, FCFS, grid2 1536 lean", dict(SFB_SP_PHASED=1, SFB_SP_SLICE=100000, SFB_SP_GRID2=1536)),
                 ):
    env(SFB_SP_PREDICT=0, **kw)
    a = timed(); check()
    b = timed(longest); check()
    extra = "  ".join("sigma %.1f: %.2f" % (sig, timed(o)) for sig, o in noisy.items())
    print("%-32s natural %.2f ms | longest first %.2f ms (%.0f QP/s) | noisy predictor %s" % (name, a, b, B / b * 1e3, extra), flush=True)
