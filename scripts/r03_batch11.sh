#!/bin/bash
# round 3, batch 11: tile kernel with the first round's element word / history value requested before the staging barrier
set -o pipefail
source scripts/r03_lib.sh
{
for w in "cfg4 --n 128" "cfg4 --n 128 --material bourdin" "cfg4 --n 96 --material miehe" "cfg5 --n 64" "quad8 --n 1000" "tri6 --n 1000" "gd3 --n 128"; do
B "$w" --workload $w || exit 1
done
} 2>&1 | tee gpurun_out/r03_batch11.txt
