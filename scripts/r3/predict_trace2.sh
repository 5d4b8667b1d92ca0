#!/bin/bash
# experiment: the second launch split into its ADMM part (grid sweep) and the polish + report part (full grid)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for G in 320 384 448 512 576 640 768; do
  OUT=$ROOT/gpurun_out/predict_split_$G
  SFB_SP_SPLIT3=1 SFB_SP_GRID3=$G rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- $B > $OUT.log 2>&1
  echo "grid3 $G: $(grep -h '"metric"' $OUT.log | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print(round(r["ms_per_step"],2), "ms")')"
  python - $OUT <<'PY'
import sys, glob, csv
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sparse" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("    sparse kernels of the last step (ms):", ["%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows[-3:]])
PY
done
