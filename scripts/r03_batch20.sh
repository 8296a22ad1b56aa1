#!/bin/bash
# round 3, batch 20: Hex8 force-vector kernel, runtime-model / general-tangent instantiations ($1 = label of the build)
set -o pipefail
source scripts/r03_lib.sh
{
B "cfg4 128 bourdin force-only ($1)" --workload cfg4 --n 128 --material bourdin --force-only || exit 1
B "cfg4 96 miehe force-only ($1)" --workload cfg4 --n 96 --material miehe --force-only || exit 1
B "cfg4 128 force-only ($1)" --workload cfg4 --n 128 --force-only || exit 1
} 2>&1 | tee -a gpurun_out/r03_batch20.txt
