#!/bin/bash
# round 3, batch 18: force-vector kernel instantiations asked for three waves per SIMD
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gradient_damage.py tests/test_gpu_emat.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests18.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests18.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg3 miehe force-only" --workload cfg3 --material miehe --force-only || exit 1
B "cfg3 force-only" --workload cfg3 --force-only || exit 1
B "cfg5g2 BASELINE" --workload cfg5g2 || exit 1
} 2>&1 | tee gpurun_out/r03_batch18.txt
