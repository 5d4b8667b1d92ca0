#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "=== nested ==="; B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | head -4
echo "=== flat ===";  SFB_MPC_STAGE=flat B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | head -4
timeout 600 python -m pytest tests/test_mpc_gpu.py -m gpu -x -q 2>&1 | tail -3
