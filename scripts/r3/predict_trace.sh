#!/bin/bash
# kernel durations of the launch in predicted order (the two qp_sparse_kernel launches + the rank kernel) for a few grids of the second launch
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for cfg in "640 27" "768 27" "640 52" "704 52" "768 52" "640 77"; do
  set -- $cfg
  OUT=$ROOT/gpurun_out/predict_trace_$1_$2
  SFB_SP_GRID3=$1 SFB_SP_PAUSE=$2 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- $B > $OUT.log 2>&1
  echo "grid3 $1 pause $2: $(grep -h '"metric"' $OUT.log | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print(round(r["ms_per_step"],2), "ms", r.get("parity_vs_oracle"))')"
  python - $OUT <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sparse" in r["Kernel_Name"] or "rank" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-3:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("    %-18s grid %6s  start %8.3f ms  duration %8.3f ms" % (r["Kernel_Name"][:18], r.get("Grid_Size_X", r.get("Grid_Size", "?")), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
done
