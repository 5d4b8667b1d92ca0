"""Closed-loop MPCSwarm ticks through the host entry point (assembly + H2D + solve + D2H): wall time per tick,
cold (first) and warm-started (later) ticks.  VARIANT/K/B from the environment."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import models_lib as M
variant = int(os.environ.get("VARIANT", 12)); K = int(os.environ.get("K", 50)); B = int(os.environ.get("B", 8192))
u0 = np.zeros((B, 2)); codes = np.zeros(B, np.int32); iters = np.zeros(B, np.uint32)
def run(ticks):
    t0 = time.perf_counter()
    rc = M.lib().sfbx_mpc_swarm_step(variant, K, C.c_double(5.0), C.c_int64(B), C.c_uint64(1), ticks,
                                      u0.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p), iters.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return time.perf_counter() - t0
run(1)
prev = 0.0
for ticks in (1, 2, 3, 4):
    t = run(ticks)
    print("ticks=%d total %.3f s  last tick %.3f s  iters of last tick: mean %.1f max %d  codes %s" % (
        ticks, t, t - prev, iters.mean(), iters.max(), np.bincount(codes, minlength=7)))
    prev = t
