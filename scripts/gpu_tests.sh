#!/bin/bash
# the whole GPU suite + a bench line (run on the GPU box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_latest.json | cut -c1-600
