cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "rccl" 2>&1 | tail -15
