import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import smooth_feedback_amd as sfb
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
for dof, ny in ((7, 3), (8, 3), (9, 3), (10, 3), (15, 6), (3, 10)):
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
    fns = {"predict": lambda: sfb.ekf_predict_batch_device(B, dof, dA.data_ptr(), dQ.data_ptr(), 0, ddt.data_ptr(), 0, dP.data_ptr(), stream=s.cuda_stream),
           "update": lambda: sfb.ekf_update_batch_device(B, dof, ny, dH.data_ptr(), dR.data_ptr(), 0, dr.data_ptr(), dP.data_ptr(), delta.data_ptr(), info.data_ptr(), stream=s.cuda_stream)}
    for name, fn in fns.items():
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3): fn()
        e1.record(s); torch.cuda.synchronize()
        print(dof, ny, name, "%.3f ms" % (e0.elapsed_time(e1) / 3))
