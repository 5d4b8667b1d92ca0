#!/bin/bash
# Per-PHASE counters of the headline launch (SFB_SP_PHASED=1 cuts it into setup / ADMM loop / polish + report kernels): what the
# waves of the setup and the polish launches spend their cycles on.  Two PMC passes, per-dispatch rows (no trace domains).
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/phase_counters
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload mpc --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --debug-knob SFB_SP_PHASED=1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for i in (1, 2, 3):
    f = glob.glob("$OUT/p%d/**/p_counter_collection.csv" % i, recursive=True)
    if not f: print("no counters in pass", i); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "qp_sparse_kernel" in r["Kernel_Name"]]
    # dispatches come in threes: setup, loop, finish (by Dispatch_Id order)
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    phase = {d: ("setup", "loop", "finish")[k % 3] for k, d in enumerate(ids)}
    for r in rows:
        acc[phase[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for ph in ("setup", "loop", "finish"):
    print("==", ph)
    for c, v in sorted(acc[ph].items()):
        print("   %-24s %16.0f   (%d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
