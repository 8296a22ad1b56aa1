#!/bin/bash
# round 3, batch 15: 2-D general-tangent tile kernels capped at 128 registers (two workgroups per CU, 12-28 bytes of scratch)
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests15.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests15.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
for w in "quad8 --n 1000 --material miehe" "quad8g2 --n 1000 --material miehe" "tri6 --n 1000 --material miehe" "tri3g3 --n 2000 --material miehe"; do
B "$w (capped at 128 VGPRs)" --workload $w || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch15.txt
