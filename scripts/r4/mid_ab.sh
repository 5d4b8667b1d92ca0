#!/bin/bash
# A/B of the packed engine's sweeps (k <= 128): block sweeps with DPP pivot broadcast (default) vs v_readlane sweeps (SFB_QP_MID_ROWS=0)
set -x
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_qp_dense_gpu.py -x -q -m gpu -k "larger_dense or padded_beyond or non_finite or explicit_workspace or max_time" 2>&1 | tail -5
BIG=1 N=400 SEED=4100 python scripts/fuzz_dense.py 2>&1 | tail -4
N=300 SEED=4200 python scripts/fuzz_dense.py 2>&1 | tail -4
SIZES=32x64,40x60,64x64,33x40 python scripts/r4/dense_iter_cost.py 2>&1 | tee gpurun_out/mid_rows.txt
SFB_QP_MID_ROWS=0 SIZES=32x64,40x60,64x64 python scripts/r4/dense_iter_cost.py 2>&1 | tee gpurun_out/mid_readlane.txt
