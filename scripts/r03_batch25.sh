#!/bin/bash
# round 3, batch 25: persistent lane-pair kernel by default (one workgroup per 16 tiles), every GET_MATRIX0 instantiation without
# off-diagonal blocks, against FA_PAIR_PERSIST=0
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gradient_damage.py tests/test_gpu_multi.py tests/test_golden.py tests/test_closed_form.py -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/r03_tests25.log 2>&1; rc=$?
tail -2 gpurun_out/r03_tests25.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
for w in "cfg3" "cfg3 --mode accumulate" "cfg3 --pattern written" "cfg3 --material miehe" "cfg3 --material bourdin" "cfg2 --n 4000" "cfg2 --n 4000 --mode accumulate" "cfg2" "cfg3micro" "cfg3micro --material bourdin" "cfg3micro --material amor-split" "gd"; do
FA_PAIR_PERSIST=0 B "$w (one workgroup per tile)" --workload $w || exit 1
B "$w (persistent)" --workload $w || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch25.txt
