#!/bin/bash
# after a change of the sparse kernel: the GPU suite, the sparse fuzz sweeps in every launch form, a bench line
cd ${GRAFT_REPO_ROOT:-.}
f() { echo "$*: $(env "$@" 2>&1 | grep -v 'tuning knob' | tail -1)"; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
f N=6000 SEED=20261101 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=4 N=4000 BMAX=48 SEED=20261102 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_GRID=3,SFB_SP_PAUSE=2 N=3000 BMAX=32 SEED=20261103 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_SP_FORCE_LAT=1 N=3000 SEED=20261104 timeout 600 python scripts/fuzz_sparse.py
f KNOBS=SFB_PLAN_UNITS=0 N=3000 SEED=20261105 timeout 600 python scripts/fuzz_sparse.py
timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_latest.json | cut -c1-300
