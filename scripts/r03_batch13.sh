#!/bin/bash
# round 3, batch 13: new defaults for the register-heavy Hex8 kernels; larger boxes for the general-tangent family
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests/test_gpu_tiles.py tests/test_gradient_damage.py tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x --timeout 300 --timeout-method thread -k "not full_size" > gpurun_out/r03_tests13.log 2>&1; rc=$?
tail -3 gpurun_out/r03_tests13.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg4 128 keepOffDiags (default)" --workload cfg4 --n 128 --keep-off-diags || exit 1
B "cfg4 128 keepOffDiags accumulate" --workload cfg4 --n 128 --keep-off-diags --mode accumulate || exit 1
B "gd3 128 (default)" --workload gd3 --n 128 || exit 1
B "cfg4 96 miehe (default 333)" --workload cfg4 --n 96 --material miehe || exit 1
for cell in 443 444; do
FA_TILE_CELL=$cell B "cfg4 96 miehe cell $cell" --workload cfg4 --n 96 --material miehe || exit 1
done
B "cfg4 96 miehe keepOffDiags (default 333)" --workload cfg4 --n 96 --material miehe --keep-off-diags || exit 1
FA_TILE_CELL=444 B "cfg4 96 miehe keepOffDiags cell 444" --workload cfg4 --n 96 --material miehe --keep-off-diags || exit 1
} 2>&1 | tee gpurun_out/r03_batch13.txt
