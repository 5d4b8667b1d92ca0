#!/bin/bash
# Counters of the LAT loop launch of the headline step (polishers off: the profiler serialises dispatches): what a LAT wave waits for.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/lat_counters
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload mpc --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --no-closed-loop --debug-knob SFB_SP_POLISHERS=0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in (1, 2, 3, 4):
    f = glob.glob("$OUT/p%d/**/p_counter_collection.csv" % i, recursive=True)
    if not f: print("no counters in pass", i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "qp_sparse_kernel" in r["Kernel_Name"]:
            k = "LAT" if "<true" in r["Kernel_Name"].replace("(bool)1", "true").replace("ILb1", "<true") or "true, false" in r["Kernel_Name"] else "STD"
            acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in sorted(acc.items()):
        for c, v in sorted(cs.items()):
            big = [x for x in v if x > 0.05 * max(v)]
            print("%-14s %-28s mean of the large dispatches %18.0f  (%d)" % (key, c, sum(big) / max(1, len(big)), len(big)))
PY
