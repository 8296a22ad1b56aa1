#!/bin/bash
# round 3, batch 23: lane-pair kernel as persistent workgroups, the next tile's lane records requested before the write-out
set -o pipefail
source scripts/r03_lib.sh
FA_PAIR_PERSIST=512 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size and not micro" > gpurun_out/r03_tests23b.log 2>&1; rc=$?
tail -2 gpurun_out/r03_tests23b.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg3 (one workgroup per tile)" --workload cfg3 || exit 1
for g in 512 1024 2048 8192; do
FA_PAIR_PERSIST=$g B "cfg3 ($g persistent workgroups)" --workload cfg3 || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch23.txt
