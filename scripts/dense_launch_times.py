"""Per-launch time distribution of the dense benchmark workload (65 536 QPs, n = 10, m = 20, bench parameters): looks for
outliers of the time-sliced ticket queue.  python scripts/dense_launch_times.py [launches]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import smooth_feedback_amd as sfb
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
w = bench.WORKLOADS["qp_dense"](sfb, 0, dev)
s = torch.cuda.current_stream()
for _ in range(2): w.step(s)
torch.cuda.synchronize()
ts = []
for _ in range(N):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); w.step(s); e1.record(s); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = np.array(ts)
print("launches", N, "ms: min %.2f median %.2f mean %.2f max %.2f" % (ts.min(), np.median(ts), ts.mean(), ts.max()))
print("sorted tail:", np.round(np.sort(ts)[-6:], 2), " outliers > 1.2 x median:", int((ts > 1.2 * np.median(ts)).sum()))
