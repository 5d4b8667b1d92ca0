#!/bin/bash
# A/B on the GPU box: the LAT loop's factor values with plain (product) or non-temporal loads (libsfb_xnt.so = -DSFB_LAT_NT=1)
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for rep in 1 2; do for L in "" smooth_feedback_amd/libsfb_xnt.so; do
  echo "lib=${L:-product}: $(SFB_LIB_PATH=$L $B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); p=r.get("parity_vs_oracle",{}); print("%.2f ms" % r["ms_per_step"], p.get("iter_mismatches"), p.get("max_abs_dx"))')"
done; done
