#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1200 python -m pytest tests/test_qp_dense_gpu.py tests/test_asif_gpu.py -m gpu -x -q 2>&1 | tail -15
