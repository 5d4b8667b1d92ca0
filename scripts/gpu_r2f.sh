#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_mpc_gpu.py tests/test_qp_sparse_gpu.py tests/test_mpc_assembly_gpu.py -m gpu -x -q 2>&1 | tail -5
for s in 100 50 200 0; do
echo "=== slice $s ==="; SFB_SP_SLICE=$s B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p
done
