#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for pad in 0 5000 9000 16000 25000; do echo "== LDS pad $pad"; SFB_QP_LDS_PAD=$pad timeout 120 python scripts/iter_cost.py 2>&1 | tail -2; done
echo "== readlane sweep"; SFB_QP_SWEEP=0 timeout 120 python scripts/iter_cost.py 2>&1 | tail -2
