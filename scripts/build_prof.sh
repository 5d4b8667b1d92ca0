#!/bin/bash
# Profiling build of libsfb.so: the sparse kernel prints the cycles it spends in the phases of the numeric
# factorisation (SFB_PROF_LDL).  Use with SFB_LIB_PATH=smooth_feedback_amd/libsfb_prof.so scripts/ldl_prof.py
# PROF_DEFS=-DSFB_SP_TIMELINE: per-item wall-clock stamps instead (scripts/timeline.py)
set -e
cd "$(dirname "$0")/../smooth_feedback_amd/csrc"
make -s
mkdir -p build_prof
cp build/*.o build_prof/
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math ${PROF_DEFS:--DSFB_PROF_LDL}"
/opt/rocm/bin/hipcc $FLAGS -c qp_sparse.hip -o build_prof/qp_sparse.o
# the same guard as the Makefile's: numbers from a build whose sweeps spill or touch in-flight registers mean nothing
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only qp_sparse.hip -o build_prof/qp_sparse.s 2> /dev/null
python3 check_sweep_spills.py build_prof/qp_sparse.s > build_prof/qp_sparse.spills || (cat build_prof/qp_sparse.spills; rm -f build_prof/qp_sparse.o; false)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfb_prof.so build_prof/*.o -Wl,-rpath,/opt/rocm/lib
