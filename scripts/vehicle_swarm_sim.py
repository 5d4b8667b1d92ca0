"""examples/mpc_asif_vehicle.cpp for a swarm: MPC (linearised, assembled and solved on the GPU) filtered by the ASI filter
(assembled and solved on the GPU), closed loop.  python scripts/vehicle_swarm_sim.py [batch] [ticks] [K_mpc] [K_asif]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples import models_lib as M

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 120
K_mpc = int(sys.argv[3]) if len(sys.argv) > 3 else 30
K_asif = int(sys.argv[4]) if len(sys.argv) > 4 else 200
r = M.vehicle_swarm_sim(batch, ticks, K_mpc, K_asif)
s = r["seconds"][2:]
print(f"{batch} vehicles, {ticks} ticks of 25 ms, MPC K = {K_mpc}, ASIF K = {K_asif}")
print(f"  per tick: MPC {1e3 * np.median(s[:, 0]):.2f} ms, ASIF {1e3 * np.median(s[:, 1]):.2f} ms (median); worst {1e3 * s.sum(1).max():.2f} ms")
print(f"  non-Optimal solves: MPC {int(r['mpc_bad'].sum())}, ASIF {int(r['asif_bad'].sum())} of {batch * ticks} each")
print(f"  smallest barrier value over the run: {r['hmin'].min():.4f} (>= 0: outside the obstacle's margin)")
d = np.abs(r["u_asif"] - r["u_mpc"]).max(axis=2)
print(f"  filter active (|u_asif - u_mpc| > 1e-3) in {np.mean(d > 1e-3):.3f} of the vehicle-ticks; u_asif range {r['u_asif'].min(axis=(0, 1))} .. {r['u_asif'].max(axis=(0, 1))}")
