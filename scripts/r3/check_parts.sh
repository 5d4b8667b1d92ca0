#!/bin/bash
# Which part of a stopping check costs a lone wave what: builds of libsfb.so that leave one part out (results differ --
# timing only), run through scripts/r3/sparse_check_cost.py.  Build here (no GPU), run on the box: BUILD=1 / RUN=1.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/smooth_feedback_amd/csrc
if [ "${BUILD:-0}" = 1 ]; then
  make -s
  for X in 1 2 4 7; do
    mkdir -p build_x$X; cp build/*.o build_x$X/
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -DSFB_CHK_EXP=$X -c qp_sparse.hip -o build_x$X/qp_sparse.o &
  done
  wait
  for X in 1 2 4 7; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfb_x$X.so build_x$X/*.o -Wl,-rpath,/opt/rocm/lib; done
fi
if [ "${RUN:-0}" = 1 ]; then
  cd $ROOT
  echo "product:"; python scripts/r3/sparse_check_cost.py 2>&1 | grep "B     1"
  for X in 1 2 4 7; do echo "without part mask $X (1 = A x rows, 2 = certificate sum, 4 = P dx rows):"; SFB_LIB_PATH=smooth_feedback_amd/libsfb_x$X.so python scripts/r3/sparse_check_cost.py 2>&1 | grep "B     1"; done
fi
