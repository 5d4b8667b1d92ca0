#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for g in 512 768 1024 1536 2048 3072; do for lw in 512 1000000000; do
echo -n "grid $g lean_waves $lw: "; SFB_SP_GRID=$g SFB_SP_LEAN_WAVES=$lw SFB_SP_SLICE=50 B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep "default params" | cut -c1-60
done; done
