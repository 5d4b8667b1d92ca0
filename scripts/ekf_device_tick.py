"""Steps/s of a swarm of vehicle EKFs resident on the GPU (EKFSwarmDevice: linearisation + state step + covariance
kernels on the device) against one host EKF<> object per filter.  python scripts/ekf_device_tick.py [batch] [steps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples import models_lib as M

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
st, P0, y = M.ekf_swarm_inputs(batch, steps, seed=1)
for name, kw in (("one-launch round, measurements resident", dict(fused=3)), ("one-launch round (Euler, 1 substep)", dict(fused=1)), ("round as 4 launches", dict(fused=2)), ("predict + update (Euler)", dict()),
                 ("predict(dt = tau/4) + update (Euler)", dict(dt=0.025)), ("predict + update (RK4)", dict(rk4=True))):
    r = M.ekf_swarm_device(st, P0, y, tau=0.1, **kw)
    s_ = np.median(r["seconds"][1:])
    where = "measurements already on the device" if kw.get("fused") == 3 else "measurement upload included"
    print(f"{name:40s} {1e3 * s_:8.3f} ms per round of {batch} filters ({where}) = {batch / s_ / 1e6:8.1f} M filter-rounds/s")
nb = 64
t0 = time.perf_counter()
M.ekf_swarm_host(st[:nb], P0[:nb], y[:, :nb], tau=0.1)
dt = (time.perf_counter() - t0) / (nb * steps)
print(f"host EKF<> object, one filter at a time: {1e6 * dt:.1f} us per predict + update = {1 / dt / 1e3:.1f} k filter-rounds/s")
