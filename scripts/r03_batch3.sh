#!/bin/bash
# round 3, batch 3: pair lanes of the Hex8 tile kernel ordered by owned node (compact stores per wave)
set -o pipefail
source scripts/r03_lib.sh
for m in 1 2 3; do
  FA_TILE_LANES=$m timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests3_$m.log 2>&1; rc=$?
  echo "FA_TILE_LANES=$m: $(tail -1 gpurun_out/r03_tests3_$m.log)"
  [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
done
{
for m in 0 1 2 3; do
FA_TILE_LANES=$m B "cfg4 128 lanes=$m"            --workload cfg4 --n 128 || exit 1
done
for m in 0 1 3; do
FA_TILE_LANES=$m B "cfg4 128 accumulate lanes=$m" --workload cfg4 --n 128 --mode accumulate || exit 1
FA_TILE_LANES=$m B "cfg4 128 keep lanes=$m"       --workload cfg4 --n 128 --keep-off-diags || exit 1
done
FA_TILE_LANES=1 B "quad8 1000 lanes=1" --workload quad8 --n 1000
FA_TILE_LANES=3 B "quad8 1000 lanes=3" --workload quad8 --n 1000
FA_TILE_LANES=0 B "quad8 1000 lanes=0" --workload quad8 --n 1000
} 2>&1 | tee gpurun_out/r03_batch3.txt
