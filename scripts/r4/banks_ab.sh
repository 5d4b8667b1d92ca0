#!/bin/bash
# A/B of the bank-aware placement of the sweep slots (sparse_plan.cpp UnitPlacer; SFB_PLAN_BANKS=0 = natural placement):
# MPC tests, headline launch time, lone-wave iteration time, and the LDS bank-conflict counters of the loop launch.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
ROOT=$PWD
{
timeout 300 python -m pytest tests/test_mpc_gpu.py -x -q -m gpu 2>&1 | tail -2
B="timeout 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
for v in 1 0 1 0; do
  echo "SFB_PLAN_BANKS=$v launch: $(SFB_PLAN_BANKS=$v $B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.2f ms" % r["ms_per_step"])')"
done
for v in 1 0; do
  echo "SFB_PLAN_BANKS=$v single agent: $(SFB_PLAN_BANKS=$v timeout 120 python scripts/single_agent_latency.py 2>/dev/null | tail -1)"
  echo "SFB_PLAN_BANKS=$v $(SFB_PLAN_BANKS=$v B=256 timeout 200 python scripts/mpc_time.py 2>/dev/null | tail -3 | tr '\n' ' ')"
done
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  OUT=$ROOT/gpurun_out/banks_pmc_$v
  SFB_PLAN_BANKS=$v timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT -o p -- \
    python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc > $OUT.log 2>&1
  python - <<PY
import csv, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "qp_sparse_kernel" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("qp_sparse_kernel")[1][:14]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, c in per.items():
    print("SFB_PLAN_BANKS=$v", k, "dispatches", len(n[k]), "LDS bank-conflict cycles per LDS instruction %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_INSTS_LDS"])),
          "conflict cycles per dispatch %.3e" % (c["SQ_LDS_BANK_CONFLICT"] / len(n[k])))
PY
done
} 2>&1 | tee gpurun_out/r4_banks_ab.txt
