#!/bin/bash
# kernel durations of the last bench step (sparse kernels), for environment settings given as arguments: "A=1 B=2" "A=3" ...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
i=0
for cfg in "$@"; do
  i=$((i+1)); OUT=$ROOT/gpurun_out/trace_last_$i
  env $cfg rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- $B > $OUT.log 2>&1
  echo "[$cfg]: $(grep -h '"metric"' $OUT.log | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print(round(r["ms_per_step"],2), "ms per step")')"
  python - $OUT <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sparse" in r["Kernel_Name"] or "rank" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 4  # warm-up + 3 steps (+ the parity run): same number of kernels each
last = rows[-n:] if n else rows
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("    %-34s grid %7s lds %6s  start %8.3f ms  duration %8.3f ms" % (r["Kernel_Name"][:34], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")),
                                                                         (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
done
