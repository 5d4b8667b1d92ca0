#!/bin/bash
# Kernel timeline of the headline step (8 192 MPC QPs, launch in predicted order): duration of every dispatch of the last step,
# for the environment given as arguments (e.g. SFB_SP_FIRST_LAT=0).  rocprofv3 --kernel-trace.
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  OUT=$ROOT/gpurun_out/launch_trace_$(echo $v | tr '= ' '__')
  rm -rf $OUT
  K=""; case $v in SFB_*) K="--debug-knob $v";; esac  # (debug knobs go through sfb_debug_set; anything else, e.g. A=1, is a plain label)
  (cd $ROOT && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --steps 3 --warmup 1 \
    --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --workload mpc $K > $OUT.log 2>&1)
  echo "== $v"
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("sfb::", "").split("(")[0][:40], int(r["Grid_Size_X"])))
rows.sort()
rows = [r for r in rows if "sparse" in r[2] or "rank" in r[2]]
# the last step: from the last setup-sized dispatch on
n = len(rows)
last = [r for r in rows[-8:]]
t0 = last[0][0]
for s, e, k, g in last:
    print("  +%8.3f ms  %8.3f ms  grid %6d  %s" % ((s - t0) / 1e6, (e - s) / 1e6, g // 64, k))
PY
done
