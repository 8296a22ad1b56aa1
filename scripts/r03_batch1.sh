#!/bin/bash
# round 3, batch 1: GPU test suite after the parity / hygiene changes, then the accumulate variants and NT stores
set -o pipefail
source scripts/r03_lib.sh
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/r03_tests1.log 2>&1; rc=$?
tail -8 gpurun_out/r03_tests1.log
# assertion failures (rc 1) do not stop the measurements; a fault, abort or time limit does
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
{
B "cfg3 overwrite"              --workload cfg3 &&
FA_LIB=$PWD/variants/nt.so B "cfg3 overwrite NT stores" --workload cfg3 &&
B "cfg3 accumulate (image init)" --workload cfg3 --mode accumulate &&
B "cfg3 accumulate written"      --workload cfg3 --mode accumulate --pattern written &&
B "cfg2 4000 accumulate"         --workload cfg2 --n 4000 --mode accumulate &&
B "cfg3 keep accumulate"         --workload cfg3 --keep-off-diags --mode accumulate &&
B "cfg4 128 overwrite"           --workload cfg4 --n 128 &&
B "cfg4 128 accumulate"          --workload cfg4 --n 128 --mode accumulate &&
B "cfg5 64 overwrite"            --workload cfg5 --n 64 &&
B "cfg5 64 accumulate"           --workload cfg5 --n 64 --mode accumulate
} 2>&1 | tee gpurun_out/r03_batch1.txt
{
B "cfg4 96 miehe (1024 thr)"         --workload cfg4 --n 96 --material miehe &&
FA_TILE_BIG=2 B "cfg4 96 miehe big LDS 512 thr" --workload cfg4 --n 96 --material miehe &&
B "cfg5 64 (1024 thr)"               --workload cfg5 --n 64 &&
FA_TILE_BIG=2 B "cfg5 64 big LDS 512 thr" --workload cfg5 --n 64
} 2>&1 | tee -a gpurun_out/r03_batch1.txt
