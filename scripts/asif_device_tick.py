"""ASIF swarm tick with the assembly on the GPU (ASIFSwarmDevice) against the host-assembled one (ASIFSwarm)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from examples import models_lib as M
B = int(os.environ.get("B", 65536)); K = int(os.environ.get("K", 10))
st, ud = M.asif_swarm_states(B, seed=0)
M.asif_swarm_device_step(st[:256], ud[:256], K)
out = M.asif_swarm_device_step(st, ud, K, ticks=4)
print("device front: B=%d K=%d: wall seconds per filter call %s; codes %s, iters mean %.1f" % (B, K, np.round(out["seconds"], 4), np.bincount(out["code"], minlength=7), out["iter"].mean()))
t0 = time.perf_counter(); h = M.asif_swarm_step(B, K, ticks=1); dt = time.perf_counter() - t0
print("host front, one tick incl. construction: %.3f s" % dt)
