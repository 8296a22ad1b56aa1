#!/bin/bash
# round 3 closing evidence: bench line + rocprofv3 kernel statistics (timed launches) + FETCH_SIZE / WRITE_SIZE passes per configuration
set -o pipefail
run() { bash scripts/r03_profiles.sh "$@" || exit 1; }
run cfg3_overwrite   cfg3:16000000:overwrite:full:K   --workload cfg3
run cfg3_accumulate  cfg3:16000000:accumulate:full:K  --workload cfg3 --mode accumulate
run cfg3_force_only  cfg3:16000000:overwrite:full:f   --workload cfg3 --force-only
run cfg4_200cubed    cfg4:8000000:overwrite:full:K    --workload cfg4
run cfg4_200cubed_force_only cfg4:8000000:overwrite:full:f --workload cfg4 --force-only
run cfg5_125x125x128_g3 cfg5:2000000:overwrite:full:K  --workload cfg5
run cfg5_125x125x128_g2 cfg5g2:2000000:overwrite:full:K --workload cfg5g2
