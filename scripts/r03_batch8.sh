#!/bin/bash
# round 3, batch 8: tile kernel with padded node lists (nodal data requested in front of the tile descriptor)
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests/test_gpu_tiles.py tests/test_gpu_multi.py tests/test_gradient_damage.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests8.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests8.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
for w in "cfg4 --n 128" "cfg4 --n 128 --mode accumulate" "cfg4 --n 96 --material miehe" "cfg5 --n 64" "cfg5g2 --n 64" "quad8 --n 1000" "tri6 --n 1000" "gd3 --n 128" "cfg4 --n 128 --force-only" "cfg5 --n 64 --force-only" "quad8 --n 1000 --force-only"; do
FA_PAD_TNODES=0 B "$w (lists not padded)" --workload $w || exit 1
B "$w (padded, early loads)" --workload $w || exit 1
done
B "cfg3 overwrite" --workload cfg3 || exit 1
B "cfg3 keepOffDiags" --workload cfg3 --keep-off-diags || exit 1
B "cfg2 4000" --workload cfg2 --n 4000 || exit 1
} 2>&1 | tee gpurun_out/r03_batch8.txt
