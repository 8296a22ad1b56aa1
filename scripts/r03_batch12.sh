#!/bin/bash
# round 3, batch 12: kernels that run one 512-thread workgroup per CU anyway (> 128 VGPRs: keepOffDiags, GradientDamage) on the
# big LDS budget with larger boxes
set -o pipefail
source scripts/r03_lib.sh
{
for w in "cfg4 --n 128 --keep-off-diags" "gd3 --n 128"; do
B "$w (default)" --workload $w || exit 1
for cell in 443 444 544; do
FA_TILE_BIG=2 FA_TILE_CELL=$cell B "$w (big LDS, 512 threads, cell $cell)" --workload $w || exit 1
done
done
} 2>&1 | tee gpurun_out/r03_batch12.txt
