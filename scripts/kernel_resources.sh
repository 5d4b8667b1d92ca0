#!/bin/bash
# Compile-time resource usage of every product kernel (hipcc -Rpass-analysis=kernel-resource-usage, gfx950):
# VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy.  No GPU needed.  Output: profiles/<tag>_kernel_resources.txt
cd "$(dirname "$0")/.."
TAG=${1:-r3}
OUT=profiles/${TAG}_kernel_resources.txt
: > $OUT
for f in smooth_feedback_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage \
    --cuda-device-only -c $f -o /dev/null 2>&1 | grep "remark:" | sed -e 's/^.*remark: //' -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | \
    awk -v file=$(basename $f) '/Function Name/{name=$3} /VGPRs:/{v=$2} /AGPRs:/{a=$2} /TotalSGPRs:/{s=$2} /ScratchSize/{sc=$3} /Occupancy/{o=$3} /LDS Size/{printf "%-20s %-110s VGPRs %3s AGPRs %3s SGPRs %3s scratch %4s B/lane  static LDS %6s B  occupancy %s waves/SIMD\n", file, name, v, a, s, sc, $4, o}' >> $OUT
done
c++filt < $OUT > $OUT.tmp && mv $OUT.tmp $OUT
wc -l $OUT
