#!/bin/bash
# Randomised parity sweeps of round 4 against the oracle (run on the GPU box; fresh seeds; every command bounded):
# dense incl. the registers-only / LDS-block engines of 32 < n+m <= 128 (MID=1) with forced tiny grids (time-sliced split launch),
# the pivoted big kernel, sparse with forced time slicing / predicted order / LAT form and both slot placements, EKF.
cd ${GRAFT_REPO_ROOT:-.}
f() { echo "$*: $(env "$@" 2>&1 | grep -v 'tuning knob' | tail -1)"; }
f N=20000 SEED=20261201 timeout 900 python scripts/fuzz_dense.py
f MID=1 N=20000 SEED=20261202 timeout 900 python scripts/fuzz_dense.py
f MID=1 SFB_MID_GRID=5 SFB_MID_SLICE=1 N=6000 SEED=20261203 timeout 900 python scripts/fuzz_dense.py
f BIG=1 N=3000 SEED=20261204 timeout 900 python scripts/fuzz_dense.py
f N=10000 SEED=20261205 timeout 600 python scripts/fuzz_sparse.py
f SFB_SP_GRID=4 N=6000 BMAX=48 SEED=20261206 timeout 600 python scripts/fuzz_sparse.py
f SFB_SP_GRID=3 SFB_SP_PAUSE=2 N=4000 BMAX=32 SEED=20261207 timeout 600 python scripts/fuzz_sparse.py
f SFB_SP_FORCE_LAT=1 N=4000 SEED=20261208 timeout 600 python scripts/fuzz_sparse.py
f SFB_PLAN_BANKS=0 N=4000 SEED=20261209 timeout 600 python scripts/fuzz_sparse.py
f N=30000 SEED=20261210 timeout 900 python scripts/fuzz_ekf.py
