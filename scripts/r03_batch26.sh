#!/bin/bash
# round 3, batch 26: force-vector kernel as persistent workgroups (ids of the next tile requested before the ordered sums)
set -o pipefail
source scripts/r03_lib.sh
FA_FVEC_PERSIST=2048 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_emat.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests26b.log 2>&1; rc=$?
tail -2 gpurun_out/r03_tests26b.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg3 force-only (one workgroup per tile)" --workload cfg3 --force-only || exit 1
for g in 2048 8192 16384; do
FA_FVEC_PERSIST=$g B "cfg3 force-only ($g persistent workgroups)" --workload cfg3 --force-only || exit 1
done
for w in "tri3 --n 2000" "cfg2 --n 4000" "cfg4 --n 128" "cfg3 --material miehe"; do
B "$w force-only (one workgroup per tile)" --workload $w --force-only || exit 1
FA_FVEC_PERSIST=4096 B "$w force-only (4096 persistent)" --workload $w --force-only || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch26.txt
