#!/bin/bash
# round 3, batch 16: 3-D elastic 512-thread tile kernels capped at 128 registers
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests16.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests16.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg5g2 64" --workload cfg5g2 --n 64 || exit 1
B "cfg5 64" --workload cfg5 --n 64 || exit 1
B "cfg5g2 64 accumulate" --workload cfg5g2 --n 64 --mode accumulate || exit 1
FA_TILE_BIG=0 B "cfg5 64 (two 512-thread workgroups)" --workload cfg5 --n 64 || exit 1
B "cfg5g2 BASELINE" --workload cfg5g2 || exit 1
B "cfg5 BASELINE" --workload cfg5 || exit 1
} 2>&1 | tee gpurun_out/r03_batch16.txt
