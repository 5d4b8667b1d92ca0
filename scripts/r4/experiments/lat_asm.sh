#!/bin/bash
# Development: assembly of the LAT instance of the sparse kernel alone (the standard instances take 3 of the 4 minutes):
#   scripts/r4/experiments/lat_asm.sh "-DSFB_SP_LAT_UNR=4" /tmp/lat.s   -> resource line + the kernel's assembly
cd "$(dirname "$0")/../../../smooth_feedback_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -DSFB_SP_LAT_ONLY $1 \
  -S --cuda-device-only -Wno-inline-asm -Wno-pass-failed -o ${2:-/tmp/lat.s} qp_sparse.hip
grep -E "^\s+\.(name|vgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):" ${2:-/tmp/lat.s} | paste - - - - - | grep lat_kernel | sed 's/_ZN3sfb20\(qp_sparse_lat_kernel\)[A-Za-z0-9_]*/\1/'
