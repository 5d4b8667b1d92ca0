"""Warm ticks of MPCSwarmDeviceLin (8 192 agents, headline model) under debug knobs: python scripts/r6/tick_knobs.py "K1=V1,K2=V2" ..."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import smooth_feedback_amd as sfb
from examples import models_lib as M

import os
batch, K, ticks = int(os.environ.get("B", 8192)), 50, 8
for spec in (sys.argv[1:] or [""]):
    pairs = sfb.debug_set_from(spec) if spec else {}
    r = M.mpc_swarm_devlin_step(12, K, batch, ticks, seed=1, want_records=False)
    ms = 1e3 * np.asarray(r["seconds"])
    print("%-40s warm ticks %s  mean of the last six %.2f ms  optimal %.4f" % (spec or "(default)", " ".join("%.2f" % v for v in ms[1:]), ms[2:].mean(), np.mean(r["code"] == 0)), flush=True)
    for k in pairs:
        sfb.debug_set(k, None)
