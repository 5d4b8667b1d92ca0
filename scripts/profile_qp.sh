#!/bin/bash
# rocprofv3 kernel-trace summary + PMC passes (separate runs) for the bench command.  Run on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -20
grep -h '"metric"' $OUT/trace.log | cut -c1-300
