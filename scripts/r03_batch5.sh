#!/bin/bash
# round 3, batch 5: interface-first launches (fa_set_interface), force-only tile sets of the tile kernel
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests/test_gpu_multi.py tests/test_gpu_tiles.py tests/test_gpu_parity.py -m gpu -q --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests5.log 2>&1; rc=$?
tail -6 gpurun_out/r03_tests5.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
for w in "cfg5 --n 64" "cfg5g2 --n 64" "quad8 --n 1000" "tri6 --n 1000" "tri3g3 --n 2000"; do
FA_TILE_FSET=0 B "$w force-only (matrix tiles)" --workload $w --force-only || exit 1
B "$w force-only (own tiles)" --workload $w --force-only || exit 1
done
FA_TILE_FSET=256 B "cfg5 --n 64 force-only (own tiles <= 256 el)" --workload cfg5 --n 64 --force-only
FA_TILE_FSET=128 B "cfg5 --n 64 force-only (own tiles <= 128 el)" --workload cfg5 --n 64 --force-only
} 2>&1 | tee gpurun_out/r03_batch5.txt
