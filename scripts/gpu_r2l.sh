#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
echo "=== lds segments (rep $rep) ==="; B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p
echo "=== no lds segments (rep $rep) ==="; SFB_PLAN_NO_LDS=1 B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p
done
rocm-smi --showclocks 2>/dev/null | head -20
