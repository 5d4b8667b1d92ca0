#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_ekf_gpu.py -m gpu -x -q 2>&1 | tail -12
timeout 300 python bench.py --workload ekf --steps 20 --warmup 3 --no-secondary 2>/dev/null | tail -1 | cut -c1-400
