#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
SFB_SP_SLICE=${SLICE:-100} SFB_LIB_PATH=$PWD/smooth_feedback_amd/libsfb_prof.so timeout 300 python scripts/timeline.py 2>&1 | grep -v amdgpu.ids
