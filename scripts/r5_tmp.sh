cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_qp_sparse_gpu.py -x -q -m gpu -k "zero_pivot" 2>&1 | tail -15
