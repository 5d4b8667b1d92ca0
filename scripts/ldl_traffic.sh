#!/bin/bash
# HBM traffic of scaling + numeric factorisation alone (max_iter = 0, no polish) for the MPC batch. GPU box.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/ldl_traffic; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  B=8192 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- python $ROOT/scripts/ldl_prof.py > $OUT/$C.log 2>&1
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/$C/**/*counter_collection.csv", recursive=True):
    tot = {}
    for r in csv.DictReader(open(f)):
        if "qp_sparse" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    print("$C", tot)
PY
done
