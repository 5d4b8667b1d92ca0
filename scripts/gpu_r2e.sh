#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== lone wave (B=1) =="; B=1 SFB_LIB_PATH=$PWD/smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep -v amdgpu.ids
echo "== B=8192 =="; B=8192 SFB_LIB_PATH=$PWD/smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep -v amdgpu.ids
bash scripts/ldl_traffic.sh 2>&1 | tail -4
