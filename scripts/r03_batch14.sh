#!/bin/bash
# round 3, batch 14: 2-D higher-order kernels of the general-tangent family (129-136 VGPRs: one 512-thread workgroup per CU) on
# the big LDS budget
set -o pipefail
source scripts/r03_lib.sh
{
for w in "quad8 --n 1000 --material miehe" "quad8g2 --n 1000 --material miehe" "tri6 --n 1000 --material miehe" "tri3g3 --n 2000 --material miehe" "quad8 --n 1000"; do
B "$w (default: 80 KB tiles)" --workload $w || exit 1
FA_TILE_BIG=2 B "$w (big LDS, 512 threads)" --workload $w || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch14.txt
