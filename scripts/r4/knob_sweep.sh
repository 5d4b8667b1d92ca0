#!/bin/bash
# headline step under launch knobs (one at a time against the defaults): is the schedule still tuned after this round's kernel changes?
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc"
for v in "A=1" "SFB_SP_LAT_LO=384" "SFB_SP_LAT_LO=512" "SFB_SP_LAT_LO=320" "SFB_SP_PAUSE=52" "SFB_SP_LEAN_WAVES=768" "SFB_SP_LEAN_WAVES=1024" "SFB_SP_LEAN_WAVES=256" \
         "SFB_SP_SLICE=25" "SFB_SP_SLICE=100" "SFB_SP_GRID=2048" "SFB_SP_GRID=1536" "SFB_SP_WAVES_PER_CU=8" "A=1"; do
  env $v timeout 300 $B 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s %9.0f QP/s  %.3f ms' % ('$v', d['value'], d['ms_per_step']))"
done
