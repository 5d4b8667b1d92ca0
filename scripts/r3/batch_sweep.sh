#!/bin/bash
# Headline model, launch time against the batch size, and the wanted-waves threshold up to which the rank kernel chooses
# the LAT form of the loop launch (SFB_SP_LAT_MAXW; default 1.25 x the waves the LAT form can hold)
cd ${GRAFT_REPO_ROOT:-.}
for B in ${BATCHES:-8192 10240 12288 16384 24576 32768}; do for MW in ${MAXW:-640 1000000}; do
echo "B=$B LAT_MAXW=$MW: $(SFB_SP_LAT_MAXW=$MW timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc --batch $B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.2f ms  %.0f QP/s" % (r["ms_per_step"], r["value"]))')"
done; done
