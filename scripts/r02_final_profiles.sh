#!/bin/bash
# round-2 closing evidence: default bench line, cfg4 at the BASELINE size (K pass and GET_INT_VECTOR pass), bench matrix
set -o pipefail
bash scripts/r02_profiles.sh cfg3_overwrite_final --workload cfg3 || exit 1
bash scripts/r02_profiles.sh cfg4_200cubed_final --workload cfg4 || exit 1
bash scripts/r02_profiles.sh cfg4_200cubed_force_only --workload cfg4 --force-only || exit 1
bash scripts/r02_bench_matrix.sh > gpurun_out/r02_matrix_final.txt 2>&1 || { tail -5 gpurun_out/r02_matrix_final.txt; exit 1; }
echo matrix done
