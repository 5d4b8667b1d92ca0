"""Closed-loop MPCSwarm ticks through the host entry point (assembly + H2D + solve + D2H): wall time per tick,
cold (first) and warm-started (later) ticks.  VARIANT/K/B from the environment."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from examples import models_lib as M
variant = int(os.environ.get("VARIANT", 12)); K = int(os.environ.get("K", 50)); B = int(os.environ.get("B", 8192))
u0 = np.zeros((B, 2)); codes = np.zeros(B, np.int32); iters = np.zeros(B, np.uint32)
DEVICE = int(os.environ.get("DEVICE", 0))   # 1: MPCSwarmDevice (records -> device assembly, device-resident warm start)
print("front:", "MPCSwarmDevice" if DEVICE else "MPCSwarm (host assembly)")
def run(ticks):
    t0 = time.perf_counter()
    fn = M.lib().sfbx_mpc_swarm_device_step if DEVICE else M.lib().sfbx_mpc_swarm_step
    rc = fn(variant, K, C.c_double(5.0), C.c_int64(B), C.c_uint64(1), ticks,
                                      u0.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p), iters.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return time.perf_counter() - t0
run(1)
ticks = 6
total = run(ticks)
secs = np.zeros(ticks); M.lib().sfbx_last_tick_seconds(secs.ctypes.data_as(C.c_void_p), ticks)
print("%d ticks, wall seconds per swarm.step(): %s  (whole call incl. construction %.3f s)" % (ticks, np.round(secs, 4), total))
print("last tick: iters mean %.1f max %d  codes %s" % (iters.mean(), iters.max(), np.bincount(codes, minlength=7)))
