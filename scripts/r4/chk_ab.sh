#!/bin/bash
# quick A/B after a sparse-kernel change: sparse + MPC GPU tests, then the headline step a few times (optionally for variant builds)
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_qp_sparse_gpu.py tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -2
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for v in A=1 ${VARIANTS} A=1 ${VARIANTS}; do
  env $v timeout 300 $B 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-50s %9.0f QP/s  %.3f ms' % ('$v', d['value'], d['ms_per_step']))"
done
