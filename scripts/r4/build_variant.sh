#!/bin/bash
# A/B builds: scripts/r4/build_variant.sh NAME "file1.hip file2.hip" "-DFOO=1 ..."  ->  smooth_feedback_amd/libsfb_NAME.so
# (use with SFB_LIB_PATH=smooth_feedback_amd/libsfb_NAME.so; build_x*/ and *.so are git-ignored but travel with gpurun)
set -e
NAME=$1; FILES=$2; DEFS=$3
cd "$(dirname "$0")/../../smooth_feedback_amd/csrc"
make -s -j8
mkdir -p build_x$NAME
cp build/*.o build_x$NAME/
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math $DEFS -c $f -o build_x$NAME/${f%.hip}.o
  if [ "$f" = qp_sparse.hip ]; then  # the Makefile's guard: a variant whose sweeps spill or touch in-flight registers is refused
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math $DEFS -S --cuda-device-only $f -o build_x$NAME/qp_sparse.s 2> /dev/null
    python3 check_sweep_spills.py build_x$NAME/qp_sparse.s > build_x$NAME/qp_sparse.spills || (cat build_x$NAME/qp_sparse.spills; rm -f build_x$NAME/qp_sparse.o; false)
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsfb_$NAME.so build_x$NAME/*.o -Wl,-rpath,/opt/rocm/lib
