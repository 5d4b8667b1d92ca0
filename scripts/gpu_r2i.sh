#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for w in 2 3 4 6; do for lw in 512 1000000000; do for s in 25 100; do
echo -n "waves/CU $w lean_waves $lw slice $s: "; SFB_SP_WAVES_PER_CU=$w SFB_SP_LEAN_WAVES=$lw SFB_SP_SLICE=$s B=8192 timeout 300 python scripts/mpc_time.py 2>&1 | grep "default params" | cut -c1-60
done; done; done
