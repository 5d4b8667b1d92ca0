#!/bin/bash
# Randomised parity sweeps of round 6 against the oracle (run on the GPU box; fresh seeds; every command bounded).  Debug knobs go
# through sfb_debug_set (KNOBS=NAME=VALUE,...): dense incl. the registers-only / LDS-block engines with forced tiny grids, the
# pivoted big kernel; sparse with forced time slicing -> launches in predicted order with the LAT loop launch, its polishers and
# helpers on four waves, the same without helpers / without polishers, the old pause point, the LAT form for whole launches,
# the standard form for the loop launch, the supernodal factorisation engine instead of the unit engine; EKF (small batches: the
# persistent kernel runs in tests/test_ekf_gpu.py and in the bench parity, at the sizes it takes).
cd ${GRAFT_REPO_ROOT:-.}
f() { echo "$*: $(env "$@" 2>&1 | grep -v 'debug knob\|amdgpu.ids' | tail -1)"; }
f N=20000 SEED=20280201 timeout 900 python scripts/fuzz_dense.py
f MID=1 N=20000 SEED=20280202 timeout 900 python scripts/fuzz_dense.py
f MID=1 KNOBS=SFB_MID_GRID=5,SFB_MID_SLICE=1 N=6000 SEED=20280203 timeout 900 python scripts/fuzz_dense.py
f BIG=1 N=3000 SEED=20280204 timeout 900 python scripts/fuzz_dense.py
f N=10000 SEED=20280205 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=4 N=6000 BMAX=48 SEED=20280206 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=4,SFB_SP_LAT_HELP=0 N=3000 BMAX=48 SEED=20280207 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=4,SFB_SP_POLISHERS=0 N=3000 BMAX=48 SEED=20280208 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=3,SFB_SP_PAUSE=27 N=4000 BMAX=32 SEED=20280209 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_FORCE_LAT=1 N=4000 SEED=20280210 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_PLAN_UNITS=0 N=4000 SEED=20280211 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=6,SFB_SP_LAT=0 N=3000 BMAX=48 SEED=20280230 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_PLAN_UNITS=0,SFB_SP_GRID=4 N=3000 BMAX=48 SEED=20280212 timeout 600 python scripts/fuzz_sparse.py
f N=30000 SEED=20280213 timeout 900 python scripts/fuzz_ekf.py
