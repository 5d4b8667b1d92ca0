#!/bin/bash
# round 5: the unit engine of the factorisation -- phase cycles (lone wave / full chip), tests, headline, setup split
cd ${GRAFT_REPO_ROOT:-.}
echo "== ldl prof, B=1"; B=1 SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep "ldl" | tail -4
echo "== ldl prof, B=8192"; B=8192 SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep "ldl" | tail -4
bash scripts/r4/chk_ab.sh
echo "== setup split"; timeout 600 python scripts/r4/setup_split.py 2>&1 | tail -8
