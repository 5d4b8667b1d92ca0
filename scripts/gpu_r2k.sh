#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
B=1 SFB_LIB_PATH=$PWD/smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/ldl_prof.py 2>&1 | grep -v amdgpu.ids
