#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for b in 1 64 256 1024 2048; do
echo "=== B $b ==="; B=$b timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 2,3p
done
