#!/bin/bash
# round 3, batch 22: tile kernel, first pair record requested in front of the barrier that ends the Gauss-point phase
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests22.log 2>&1; rc=$?
tail -2 gpurun_out/r03_tests22.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
for w in "cfg4 --n 128" "cfg4 --n 128 --mode accumulate" "cfg4 --n 128 --keep-off-diags" "cfg4 --n 96 --material miehe" "cfg5 --n 64" "cfg5g2 --n 64" "quad8 --n 1000" "tri6 --n 1000" "gd3 --n 128"; do
B "$w" --workload $w || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch22.txt
