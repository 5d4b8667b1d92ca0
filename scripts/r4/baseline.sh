#!/bin/bash
# baseline of the dense mid sizes with the current build: per-iteration cost (lone wave / batch), setup, and the size table at 65 536
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
SIZES=16x32,20x40,32x32,32x64,40x60,64x64 python scripts/r4/dense_iter_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_base_iter.txt
B=${B:-65536} SIZES=16x32,20x40,32x32,32x64,40x60,64x64 OUT=gpurun_out/r4_base_sizes.json python scripts/r3/dense_sizes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_base_sizes.txt
