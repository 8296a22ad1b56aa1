#!/bin/bash
# round 3, batch 6: where the force-vector targets stand; accumulate with the 512-thread big-LDS geometry on cfg4 / AmorPhase
set -o pipefail
source scripts/r03_lib.sh
{
B "gd cfg3 mesh K" --workload gd || exit 1
B "gd cfg3 mesh force-only" --workload gd --force-only || exit 1
B "gd3 128 K" --workload gd3 --n 128 || exit 1
B "gd3 128 force-only" --workload gd3 --n 128 --force-only || exit 1
B "cfg5 64 force-only" --workload cfg5 --n 64 --force-only || exit 1
B "cfg4 128 overwrite" --workload cfg4 --n 128 || exit 1
B "cfg4 128 accumulate" --workload cfg4 --n 128 --mode accumulate || exit 1
FA_TILE_BIG=2 B "cfg4 128 overwrite big512" --workload cfg4 --n 128 || exit 1
FA_TILE_BIG=2 B "cfg4 128 accumulate big512" --workload cfg4 --n 128 --mode accumulate || exit 1
} 2>&1 | tee gpurun_out/r03_batch6.txt
