#!/bin/bash
# A/B on the GPU box: the first check interval of a launch in predicted order in the LAT form (default) against the fused first launch
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_qp_sparse_gpu.py tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for v in "A=1" "SFB_SP_FIRST_LAT=0" "A=1" "SFB_SP_FIRST_LAT=0"; do
  echo "== $v"
  env $v timeout 300 $B 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('parity_vs_oracle'))"
done
