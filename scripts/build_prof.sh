#!/bin/bash
# Profiling build of libsfb.so: the sparse kernel prints the cycles it spends in the phases of the numeric
# factorisation (SFB_PROF_LDL).  Use with SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so scripts/ldl_prof.py
# PROF_DEFS=-DSFB_SP_TIMELINE: per-item wall-clock stamps instead (scripts/timeline.py)
set -e
cd "$(dirname "$0")/../smooth_feedback_amd/csrc"
make -s
mkdir -p build_prof
cp build/*.o build_prof/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math ${PROF_DEFS:--DSFB_PROF_LDL} -c qp_sparse.hip -o build_prof/qp_sparse.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfb_prof.so build_prof/*.o -Wl,-rpath,/opt/rocm/lib
