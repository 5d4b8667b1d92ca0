#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for mode in -1 1000000000; do
for b in 256 384 512 640 768 1024 1536; do
echo "=== lean_waves $mode B $b ==="; SFB_SP_LEAN_WAVES=$mode B=$b timeout 300 python scripts/mpc_time.py 2>&1 | grep -v amdgpu.ids | sed -n 3,3p
done; done
