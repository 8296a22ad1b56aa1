#!/bin/bash
# round 3, batch 21: cfg4 default kernel (121 VGPRs, two workgroups per CU, 3^3 boxes) against one workgroup on the whole LDS
set -o pipefail
source scripts/r03_lib.sh
{
B "cfg4 128 (default)" --workload cfg4 --n 128 || exit 1
for cell in 444 443 544; do
FA_TILE_BIG=2 FA_TILE_CELL=$cell B "cfg4 128 (big LDS, 512 threads, cell $cell)" --workload cfg4 --n 128 || exit 1
FA_TILE_BIG=1 FA_TILE_CELL=$cell B "cfg4 128 (big LDS, 1024 threads, cell $cell)" --workload cfg4 --n 128 || exit 1
done
FA_TILE_BIG=1 FA_TILE_CELL=444 B "cfg4 128 accumulate (big LDS, 1024 threads, cell 444)" --workload cfg4 --n 128 --mode accumulate || exit 1
} 2>&1 | tee gpurun_out/r03_batch21.txt
