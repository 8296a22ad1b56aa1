#!/bin/bash
# copies what scripts/r03_evidence.sh left under gpurun_out/r03_prof/ into profiles/ and merges the traffic figures
for t in cfg3_overwrite cfg3_accumulate cfg3_force_only cfg4_200cubed cfg4_200cubed_force_only cfg5_125x125x128_g3 cfg5_125x125x128_g2; do
  d=gpurun_out/r03_prof/$t
  cp $d/bench.json profiles/r03_${t}_bench.json; cp $d/timed_kernel_stats.csv profiles/r03_${t}_kernel_stats.csv
  cp $(ls -t $d/stats/vm/*_kernel_stats.csv | head -1) profiles/r03_${t}_rocprof_stats_all_launches.csv; cp $d/traffic.txt profiles/r03_${t}_traffic.txt
done
python3 scripts/make_traffic.py --merge gpurun_out/r03_prof/cfg*/traffic_entry.json
sed -i 's#"gpurun_out/r03_prof/\([a-z0-9_x]*\)"#"profiles/r03_\1_{bench.json,kernel_stats.csv,traffic.txt} (gpurun_out/r03_prof/\1)"#' profiles/traffic.json
