#!/bin/bash
# A/B of library builds on ONE box: alternates the MPC bench between the .so files given as arguments
# (paths relative to the repo root), ROUNDS times each.  Prints ms per step of every run.
ROOT=${GRAFT_REPO_ROOT:-$PWD}; ROUNDS=${ROUNDS:-3}; WL=${WL:-mpc}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    ms=$(SFB_LIB_PATH=$ROOT/$lib python $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined 2>/dev/null | tail -1 | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r $lib $ms ms"
  done
done
