#!/bin/bash
# quick parity + per-iteration cost of the packed engine (k <= 128)
cd "$(dirname "$0")/../.."
python -m pytest tests/test_qp_dense_gpu.py -x -q -m gpu -k "larger_dense or padded_beyond or non_finite or explicit_workspace" 2>&1 | tail -2
BIG=1 N=${N:-300} SEED=${SEED:-5100} python scripts/fuzz_dense.py 2>&1 | tail -3
SIZES=${SIZES:-32x64,40x60,64x64} python scripts/r4/dense_iter_cost.py 2>&1 | grep -v amdgpu.ids
