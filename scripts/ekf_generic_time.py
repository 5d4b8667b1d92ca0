"""Throughput of the generic (one filter per wavefront) EKF kernel at the sizes the reference's tests use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smooth_feedback_amd as sfb
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for dof, ny in ((6, 3), (6, 6), (4, 4), (7, 3), (8, 3), (9, 3), (10, 3), (3, 10), (12, 6), (16, 16)):
    B = 1 << 18
    def spd(n):
        M = rng.standard_normal((n, n)); return M @ M.T + n * np.eye(n)
    P = np.tile(spd(dof).ravel(), (B, 1)); A = np.tile(0.1 * rng.standard_normal((dof, dof)).ravel(), (B, 1))
    Q = np.tile(0.01 * spd(dof).ravel(), (B, 1)); H = np.tile(rng.standard_normal((ny, dof)).T.ravel(), (B, 1))
    R = np.tile(spd(ny).ravel(), (B, 1)); r = rng.standard_normal((B, ny)); dt = np.full(B, 1e-2)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dP, dA, dQ, dH, dR, dr, ddt = [T(a) for a in (P, A, Q, H, R, r, dt)]
    delta = torch.empty((B, dof), dtype=torch.float64, device=dev); info = torch.empty(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream()
    def run():
        sfb.ekf_predict_update_batch_device(B, dof, ny, dA.data_ptr(), dQ.data_ptr(), 0, ddt.data_ptr(), 0, dH.data_ptr(), dR.data_ptr(),
                                            0, dr.data_ptr(), dP.data_ptr(), delta.data_ptr(), info.data_ptr(), stream=s.cuda_stream)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): run()
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byt = 8 * (3 * dof * dof + ny * dof + ny * ny + ny + 1) + 8 * (dof * dof + dof)
    print("dof %2d ny %2d: %.3f ms for %d filters -> %.1f M steps/s, %.0f GB/s of %d B per filter" % (dof, ny, ms, B, B / ms / 1e3, B * byt / ms / 1e6, byt))
