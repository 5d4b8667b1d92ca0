#!/bin/bash
# The last randomised parity sweeps of round 3 (run on the GPU box); every command bounded.
cd ${GRAFT_REPO_ROOT:-.}
f() { echo "$*: $(env "$@" 2>&1 | grep -v 'tuning knob' | tail -1)"; }
f N=30000 SEED=20260929 timeout 900 python scripts/fuzz_dense.py
f BIG=1 N=4000 SEED=20260930 timeout 900 python scripts/fuzz_dense.py
f N=15000 SEED=20260929 timeout 600 python scripts/fuzz_sparse.py
f SFB_SP_GRID=4 N=10000 BMAX=48 SEED=20260931 timeout 600 python scripts/fuzz_sparse.py
f SFB_SP_GRID=3 SFB_SP_PAUSE=2 N=5000 BMAX=32 SEED=20260932 timeout 600 python scripts/fuzz_sparse.py
f SFB_SP_FORCE_LAT=1 N=5000 SEED=20260933 timeout 600 python scripts/fuzz_sparse.py
f N=50000 SEED=20260929 timeout 900 python scripts/fuzz_ekf.py
