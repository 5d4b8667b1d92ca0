#!/bin/bash
# A/B: LDS budget of a sparse item (plan side, SFB_PLAN_LDS doubles) -> resident waves per CU in the standard-form launches
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for v in "A=1" "SFB_PLAN_LDS=1664" "SFB_PLAN_LDS=1600" "SFB_PLAN_LDS=1536" "SFB_PLAN_LDS=1792" "A=1"; do
  env $v timeout 300 $B 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s %9.0f QP/s  %.3f ms  parity %s' % ('$v', d['value'], d['ms_per_step'], d.get('parity_vs_oracle')))"
done
