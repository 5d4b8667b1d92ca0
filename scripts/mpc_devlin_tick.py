"""Tick time of the MPC swarm by where the linearisation runs: MPCSwarmDevice (host threads + packed upload) against
MPCSwarmDeviceLin (one GPU thread per agent and node).  python scripts/mpc_devlin_tick.py [batch] [K] [ticks]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples import models_lib as M

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
for variant in (12, 6):
    M.mpc_swarm_step(variant, K, batch, ticks, seed=1, device=True)
    host = M.last_tick_seconds(ticks) if hasattr(M, "last_tick_seconds") else None
    r = M.mpc_swarm_devlin_step(variant, K, batch, ticks, seed=1, want_records=False)
    print(f"variant {variant} K {K} batch {batch}: ms per tick")
    if host is not None:
        print("  host linearisation  ", " ".join(f"{1e3 * s:7.2f}" for s in host))
    print("  device linearisation", " ".join(f"{1e3 * s:7.2f}" for s in r["seconds"]), f" packed={r['packed']} optimal={np.mean(r['code'] == 0):.4f}")
