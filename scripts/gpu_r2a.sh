#!/bin/bash
# round 2, first GPU loop: pruned-plan parity + timing with / without pruning
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mpc_gpu.py tests/test_qp_sparse_gpu.py -m gpu -x -q 2>&1 | tail -15
echo "=== pruned ==="; B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | tail -12
echo "=== whole pattern ==="; NO_PRUNE=1 B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | tail -12
echo "=== bench ==="; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_r2a.json | cut -c1-3000
