#!/bin/bash
# round 3, batch 7: lane-pair kernel with the gathers issued in front of the image initialisation
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests7.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests7.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg3 overwrite" --workload cfg3 || exit 1
B "cfg3 overwrite (again)" --workload cfg3 || exit 1
B "cfg3 accumulate" --workload cfg3 --mode accumulate || exit 1
B "cfg3 keepOffDiags" --workload cfg3 --keep-off-diags || exit 1
B "cfg3 miehe" --workload cfg3 --material miehe || exit 1
B "cfg2 4000" --workload cfg2 --n 4000 || exit 1
B "cfg2 4000 accumulate" --workload cfg2 --n 4000 --mode accumulate || exit 1
B "cfg3 force-only" --workload cfg3 --force-only || exit 1
} 2>&1 | tee gpurun_out/r03_batch7.txt
