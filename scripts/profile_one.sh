#!/bin/bash
# rocprofv3 summaries for ONE workload (kernel trace + separate PMC passes). Run on the GPU box: profile_one.sh <tag> <workload>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; WL=$2
OUT=$ROOT/gpurun_out/prof_${TAG}_$WL
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
grep -h '"metric"' $OUT/trace.log | cut -c1-200
