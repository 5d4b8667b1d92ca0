cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
for K in "SFB_SP_LAT_HELP=1" "SFB_SP_LAT_HELP=0"; do
OUT=$ROOT/gpurun_out/pipe_trace_$K
rm -rf $OUT
(cd $ROOT && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-closed-loop --workload mpc --debug-knob $K > $OUT.log 2>&1)
echo "== $K"
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/**/t_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("sfb::", "").split("(")[0][:40], int(r["Grid_Size_X"]), r.get("Queue_Id"), r.get("Stream_Id")))
rows.sort()
rows = [r for r in rows if "sparse" in r[2] or "rank" in r[2]]
last = rows[-44:]
t0 = last[0][0]
for s, e, k, g, q, st in last:
    if e - s > 50000: print("  +%8.3f ms  %8.3f ms  grid %6d  q %s st %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, g // 64, q, st, k))
d = json.loads(open("$OUT.log").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["pipelined"])
PY
done
