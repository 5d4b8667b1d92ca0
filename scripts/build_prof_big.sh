#!/bin/bash
# Profiling build of libsfb.so for the big dense kernel: item 0 prints the time its sweeps spend per phase.
set -e
cd "$(dirname "$0")/../smooth_feedback_amd/csrc"
make -s
mkdir -p build_prof
cp build/*.o build_prof/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -w -DSFB_BIG_PROF -c qp_dense_big.hip -o build_prof/qp_dense_big.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfb_prof.so build_prof/*.o -Wl,-rpath,/opt/rocm/lib
