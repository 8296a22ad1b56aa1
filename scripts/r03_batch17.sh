#!/bin/bash
# round 3, batch 17: Hex20 Gauss2^3 (134 VGPRs: one 512-thread workgroup per CU on 80 KB tiles today) on the whole LDS
set -o pipefail
source scripts/r03_lib.sh
{
B "cfg5g2 64 (default)" --workload cfg5g2 --n 64 || exit 1
FA_TILE_BIG=2 B "cfg5g2 64 (big LDS, 512 threads)" --workload cfg5g2 --n 64 || exit 1
FA_TILE_BIG=1 B "cfg5g2 64 (big LDS, 1024 threads)" --workload cfg5g2 --n 64 || exit 1
FA_TILE_BUDGET=120000 FA_TILE_BIG=2 B "cfg5g2 64 (120 KB, 512 threads)" --workload cfg5g2 --n 64 || exit 1
B "cfg5g2 64 accumulate (default)" --workload cfg5g2 --n 64 --mode accumulate || exit 1
FA_TILE_BIG=2 B "cfg5g2 64 accumulate (big LDS, 512 threads)" --workload cfg5g2 --n 64 --mode accumulate || exit 1
} 2>&1 | tee gpurun_out/r03_batch17.txt
