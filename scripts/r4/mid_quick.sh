#!/bin/bash
# quick parity + per-iteration cost of the on-chip kernel for 32 < n + m <= 128 (qp_dense_mid.hip)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_qp_dense_gpu.py -x -q -m gpu 2>&1 | tail -4
MID=1 N=${N:-300} SEED=${SEED:-7100} python scripts/fuzz_dense.py 2>&1 | tail -6
SIZES=${SIZES:-16x32,20x40,32x32,32x64,40x60,64x64} python scripts/r4/dense_iter_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_mid_iter.txt
