cd ${GRAFT_REPO_ROOT:-.}
timeout 600 python -m pytest tests/test_mpc_gpu.py -x -q -m gpu -k "resident" 2>&1 | tail -5
