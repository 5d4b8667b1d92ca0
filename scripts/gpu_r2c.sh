#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for w in 12 10 8 6 4 3; do
echo "=== waves/CU $w ==="; SFB_SP_WAVES_PER_CU=$w B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p
done
