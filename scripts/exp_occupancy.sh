#!/bin/bash
# A/B occupancy experiment for the dense QP kernel (run on the GPU box from the repo root).
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $1"; shift; env "$@" timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined 2>&1 | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
run base X=1
run pad1wave SFB_QP_LDS_PAD=25000
for v in "$@"; do run $v SFB_LIB_PATH=$PWD/smooth_feedback_amd/libsfb_$v.so; done
