#!/bin/bash
# round 3, batch 2: row-image write-out of the 3-D elastic tile kernels (Hex20): correctness, then timings
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread -k "high_order or tiles or hex20 or Hex20 or cfg5 or full_size or elastic or micro_fracture_laws or accumulate" > gpurun_out/r03_tests2.log 2>&1; rc=$?
tail -6 gpurun_out/r03_tests2.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
FA_TILE_IMG=0 B "cfg5 64 g3 legacy"              --workload cfg5 --n 64 &&
B "cfg5 64 g3 image"                              --workload cfg5 --n 64 &&
FA_TILE_BUDGET=140000 B "cfg5 64 g3 image b140k"  --workload cfg5 --n 64 &&
FA_TILE_SYM=0 FA_TILE_BUDGET=110000 B "cfg5 64 g3 image nosym b110k" --workload cfg5 --n 64 &&
FA_TILE_IMG=0 B "cfg5 64 g2 legacy"              --workload cfg5g2 --n 64 &&
B "cfg5 64 g2 image"                              --workload cfg5g2 --n 64 &&
FA_TILE_BUDGET=70000 B "cfg5 64 g2 image b70k"    --workload cfg5g2 --n 64 &&
FA_TILE_SYM=0 FA_TILE_BUDGET=50000 B "cfg5 64 g2 image nosym b50k" --workload cfg5g2 --n 64 &&
B "cfg5 64 g3 image accumulate"                   --workload cfg5 --n 64 --mode accumulate &&
B "cfg3 accumulate (batched RMW)"                 --workload cfg3 --mode accumulate &&
B "cfg2 4000 accumulate"                          --workload cfg2 --n 4000 --mode accumulate &&
B "cfg4 96 miehe"                                 --workload cfg4 --n 96 --material miehe
} 2>&1 | tee gpurun_out/r03_batch2.txt
