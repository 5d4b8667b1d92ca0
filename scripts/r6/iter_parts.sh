#!/bin/bash
# Which part of an ADMM iteration of the LAT form costs what, alone and with the chip full (three LAT waves per CU): builds of
# libsfb.so that leave a part out (results are garbage -- timing only).  BUILD=1 here (no GPU), RUN=1 on the box.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/smooth_feedback_amd/csrc
if [ "${BUILD:-0}" = 1 ]; then
  make -s
  for X in 1 2; do
    mkdir -p build_x$X; cp build/*.o build_x$X/
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -DSFB_ITER_EXP=$X -c qp_sparse.hip -o build_x$X/qp_sparse.o &
  done
  wait
  for X in 1 2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfb_x$X.so build_x$X/*.o -Wl,-rpath,/opt/rocm/lib; done
fi
if [ "${RUN:-0}" = 1 ]; then
  cd $ROOT
  echo "product:"; python scripts/r6/iter_parts.py 2>&1 | grep "^B"
  echo "without the update phases (sweeps + D only):"; SFB_LIB_PATH=smooth_feedback_amd/libsfb_x1.so python scripts/r6/iter_parts.py 2>&1 | grep "^B"
  echo "without the sweeps (update phases only):"; SFB_LIB_PATH=smooth_feedback_amd/libsfb_x2.so python scripts/r6/iter_parts.py 2>&1 | grep "^B"
fi
