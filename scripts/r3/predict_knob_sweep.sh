#!/bin/bash
# Zero-code sweeps of the launch in predicted order on the headline batch (run on the GPU box): pause point of the first
# launch, its grid / slice / load flavour, waves of the LAT loop launch.  ms per launch; results are bit-identical by test.
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc"
run() { echo "$*: $(env "$@" $B 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.2f ms" % r["ms_per_step"])')"; }
run SFB_SP_PAUSE=27
run SFB_SP_PAUSE=52
run SFB_SP_PAUSE=77
run SFB_SP_SLICE=25
run SFB_SP_SLICE=100
run SFB_SP_LEAN_WAVES=256
run SFB_SP_LEAN_WAVES=1024
run SFB_SP_LEAN_WAVES=100000
run SFB_SP_GRID=2048
run SFB_SP_GRID=2304
run SFB_SP_WAVES_PER_CU=8
run SFB_SP_WAVES_PER_CU=9
run SFB_SP_GRID3=320
run SFB_SP_GRID3=384
run SFB_SP_GRID3=448
run SFB_SP_GRID3=512
run SFB_SP_LAT=0
