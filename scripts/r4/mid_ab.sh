#!/bin/bash
# A/B of builds of the on-chip kernel for 32 < n + m <= 128: per-iteration cost, lone wave and batch
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for v in "" ${VARIANTS}; do
  lib=smooth_feedback_amd/libsfb${v:+_$v}.so
  echo "== ${v:-default} ($lib)"
  SFB_LIB_PATH=$lib B=${B:-16384} SIZES=${SIZES:-16x32,20x40,32x64,40x60,64x64} python scripts/r4/dense_iter_cost.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_mid_ab.txt
