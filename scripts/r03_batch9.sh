#!/bin/bash
# round 3, batch 9: FA_ACCUMULATE of the lane-pair kernel from an LDS image initialised with K (FA_PAIR_ACC=1 build), now with
# the gathers issued in front of the image loads
set -o pipefail
source scripts/r03_lib.sh
FA_LIB=$PWD/variants/acc.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread -k "accumul" > gpurun_out/r03_tests9.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests9.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg3 accumulate (RMW stream-out)" --workload cfg3 --mode accumulate || exit 1
FA_LIB=$PWD/variants/acc.so B "cfg3 accumulate (image from K, early gathers)" --workload cfg3 --mode accumulate || exit 1
B "cfg2 4000 accumulate (RMW stream-out)" --workload cfg2 --n 4000 --mode accumulate || exit 1
FA_LIB=$PWD/variants/acc.so B "cfg2 4000 accumulate (image from K)" --workload cfg2 --n 4000 --mode accumulate || exit 1
B "cfg3 miehe accumulate (RMW stream-out)" --workload cfg3 --material miehe --mode accumulate || exit 1
FA_LIB=$PWD/variants/acc.so B "cfg3 miehe accumulate (image from K)" --workload cfg3 --material miehe --mode accumulate || exit 1
} 2>&1 | tee gpurun_out/r03_batch9.txt
