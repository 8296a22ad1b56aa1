#!/bin/bash
# round 3, batch 4: what do the K stores of the tile kernels cost?  Default build against the no-store ablation build
set -o pipefail
source scripts/r03_lib.sh
{
for m in 0 1; do
FA_TILE_LANES=$m B "cfg4 128 lanes=$m"                                --workload cfg4 --n 128 || exit 1
FA_TILE_LANES=$m FA_LIB=$PWD/variants/nw.so B "cfg4 128 lanes=$m no stores" --workload cfg4 --n 128 || exit 1
done
B "cfg5 64 g3"                                   --workload cfg5 --n 64 || exit 1
FA_LIB=$PWD/variants/nw.so B "cfg5 64 g3 no stores" --workload cfg5 --n 64 || exit 1
B "cfg5 64 g2"                                   --workload cfg5g2 --n 64 || exit 1
FA_LIB=$PWD/variants/nw.so B "cfg5 64 g2 no stores" --workload cfg5g2 --n 64 || exit 1
B "cfg4 128 force-only" --workload cfg4 --n 128 --force-only
FA_FVEC=0 B "cfg4 128 force-only tile kernel (phases 0-1)" --workload cfg4 --n 128 --force-only
} 2>&1 | tee gpurun_out/r03_batch4.txt
