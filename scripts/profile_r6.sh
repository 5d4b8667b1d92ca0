#!/bin/bash
# Round-6 rocprofv3 evidence, run on the GPU box:  bash scripts/profile_r6.sh [workloads...]   (default: all)
# Per workload: kernel trace + stats, FETCH_SIZE / WRITE_SIZE (separate passes), and SQ / TCC counter passes
# (never combined with trace domains other than --kernel-trace).  Output under gpurun_out/prof_r6_*; the summaries are
# copied into profiles/r6_* by `python scripts/summarize_profiles_r3.py r6`.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
run_passes() {  # <name> <command...>
  local NAME=$1; shift
  local OUT=$ROOT/gpurun_out/prof_r6_$NAME
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
  local i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
              "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
              "SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" \
              "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES_EQ_64"; do
    i=$((i+1))
    # (counter passes serialise the dispatches: the polishers, which run NEXT TO the loop launch and wait for its end, are switched
    #  off for them -- same work, done by the finish launch instead)
    timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -o p -- "$@" $PMC_EXTRA > $OUT/pmc$i.log 2>&1 || echo "pass $i ($CTRS) failed" >> $OUT/failed.txt
  done
  grep -h '"metric"' $OUT/trace.log | cut -c1-160
}
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary"
WL=${@:-mpc ekf qp_dense dense_mid}
for w in $WL; do
  case $w in
    mpc) PMC_EXTRA="--debug-knob SFB_SP_POLISHERS=0" run_passes $w $B --workload $w ;;
    ekf|qp_dense) PMC_EXTRA="" run_passes $w $B --workload $w ;;
    dense_mid) PMC_EXTRA="" run_passes dense_mid python $ROOT/scripts/r4/dense_mid_profile.py ;;
  esac
done
echo done
