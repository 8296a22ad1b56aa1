#!/bin/bash
# round 3, batch 10: force-vector kernel with node / element lists padded per tile (loads in front of the descriptor)
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests10.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests10.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
for w in "cfg3" "tri3 --n 2000" "cfg4 --n 128" "gd" "gd3 --n 128" "cfg3 --material miehe" "cfg2 --n 4000"; do
FA_PAD_TNODES=0 B "$w force-only (lists packed)" --workload $w --force-only || exit 1
B "$w force-only (lists padded, early loads)" --workload $w --force-only || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch10.txt
