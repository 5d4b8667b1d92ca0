#!/bin/bash
# builds the micro-benchmarks under scripts/ubench/ into scripts/ubench/bin/ (git-ignored); run them on the GPU box
cd "$(dirname "$0")/ubench" && mkdir -p bin
python3 gen_sweep_units.py > sweep_units_asm.h || exit 1  # (variants of the sparse sweep unit, sweep_units.hip)
for s in lat thr ldschain issue mix sweep_units; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -w $s.hip -o bin/$s || exit 1; done
