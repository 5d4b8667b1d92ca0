#!/bin/bash
# round 5, first look: factorisation phase cycles of a lone wave and of a full chip, the setup / polish split, a headline line
cd ${GRAFT_REPO_ROOT:-.}
echo "== ldl prof, B=1"; B=1 SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep -v "tuning knob" | tail -4
echo "== ldl prof, B=8192"; B=8192 SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep -v "tuning knob" | tail -4
echo "== setup split"; timeout 600 python scripts/r4/setup_split.py 2>&1 | tail -9
echo "== bench"; timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipelined --no-secondary --workload mpc 2>&1 | tail -1 | cut -c1-1500
