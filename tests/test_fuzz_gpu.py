"""Randomised parity sweeps (scripts/fuzz_dense.py, fuzz_sparse.py, fuzz_ekf.py) as part of the GPU suite: a few hundred random
configurations each -- sizes across all dense kernels, random patterns and pruned plans with violating items, solver
parameters incl. max_iter 0 / 1, stop_check_iter 0 / 1, infinite and equal bounds, warm starts -- every one bit-identical
to the CPU oracle (codes, iteration counts, primal, dual, objective).  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,n", [("fuzz_dense.py", 400), ("fuzz_sparse.py", 150), ("fuzz_ekf.py", 300)])
def test_randomised_parity_sweep(script, n):
    env = dict(os.environ, N=str(n), SEED="424242")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], env=env, cwd=os.path.join(ROOT, "scripts"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "%d configurations, 0 mismatching" % n in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("script,n,knobs", [
    ("fuzz_sparse.py", 150, {"KNOBS": "SFB_SP_GRID=4", "BMAX": "40"}),            # tiny grid: time slicing + launch in predicted order (LAT loop launch)
    ("fuzz_sparse.py", 100, {"KNOBS": "SFB_SP_GRID=3,SFB_SP_PAUSE=2", "BMAX": "24"}),
    ("fuzz_sparse.py", 100, {"KNOBS": "SFB_SP_FORCE_LAT=1"}),                      # the LAT form (chained sweeps, vectors in LDS) for whole launches
    ("fuzz_sparse.py", 100, {"KNOBS": "SFB_SP_GRID=4,SFB_SP_LAT=0", "BMAX": "40"}),  # predicted order with the standard-form loop launch
    ("fuzz_sparse.py", 150, {"KNOBS": "SFB_PLAN_UNITS=0"}),                        # the supernodal engine of the numeric factorisation for every plan
    ("fuzz_sparse.py", 100, {"KNOBS": "SFB_PLAN_UNITS=0,SFB_SP_GRID=4", "BMAX": "40"}),
])
def test_randomised_parity_sweep_of_the_other_launch_shapes(script, n, knobs):
    """The same sweeps with the debug knobs that select the non-default engines / launch shapes (the scripts hand KNOBS to
    sfb_debug_set): results never depend on them."""
    env = dict(os.environ, N=str(n), SEED="31337", **knobs)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)], env=env, cwd=os.path.join(ROOT, "scripts"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "%d configurations, 0 mismatching" % n in out.stdout, out.stdout[-2000:]
