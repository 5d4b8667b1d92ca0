cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
for W in 192 0; do
SFB_SP_PREPOLISH=$W rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$W -o t -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc > /tmp/tr$W.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/tr$W/**/t_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "sfb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-6]["Start_Timestamp"]) if len(rows) >= 6 else int(rows[0]["Start_Timestamp"])
print("PREPOLISH=$W  (last step)")
for r in rows[-6:]:
    n = r["Kernel_Name"].split("(")[0].replace("void sfb::","")[:40]
    print("  %-42s start %8.3f ms  end %8.3f ms  dur %7.3f ms" % (n, (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
done
