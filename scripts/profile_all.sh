#!/bin/bash
# rocprofv3 summaries for the three workloads (kernel trace + separate PMC passes). Run on the GPU box: profile_all.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for WL in mpc ekf qp_dense; do bash $ROOT/scripts/profile_one.sh $1 $WL; done
