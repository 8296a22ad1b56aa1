#!/bin/bash
# round 3, batch 24: persistent lane-pair kernel, number of workgroups
set -o pipefail
source scripts/r03_lib.sh
{
for g in 4096 8192 16384 32768 65536 131072; do
FA_PAIR_PERSIST=$g B "cfg3 ($g persistent workgroups)" --workload cfg3 || exit 1
done
FA_PAIR_PERSIST=8192 B "cfg3 2000^2 (8192)" --workload cfg3 --n 2000 || exit 1
B "cfg3 2000^2 (one workgroup per tile)" --workload cfg3 --n 2000 || exit 1
FA_PAIR_PERSIST=8192 B "cfg3 1000^2 (8192)" --workload cfg3 --n 1000 || exit 1
B "cfg3 1000^2 (one workgroup per tile)" --workload cfg3 --n 1000 || exit 1
} 2>&1 | tee gpurun_out/r03_batch24.txt
