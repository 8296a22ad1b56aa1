#!/bin/bash
# cfg4 / cfg5 at their BASELINE sizes: one JSON line each + rocprofv3 kernel stats of the same command
export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out/r02_cfg45
for w in "cfg4" "cfg5" "cfg5g2"; do
  echo "== $w"; date +%T
  timeout -k 10 500 python bench.py --workload $w --cpu-elems 0 --steps 5 --warmup 2 > gpurun_out/r02_cfg45/$w.json 2> gpurun_out/r02_cfg45/$w.err || { echo "FAILED $w"; tail -5 gpurun_out/r02_cfg45/$w.err; exit 1; }
  cat gpurun_out/r02_cfg45/$w.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs nnz %d' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds'], d['config']['nnz_per_gpu']))"
done
for w in "cfg4" "cfg5"; do
  echo "== rocprof $w"; date +%T
  (cd /tmp && timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_cfg45/prof_$w -- python3 $R/bench.py --workload $w --cpu-elems 0 --steps 3 --warmup 1 > $R/gpurun_out/r02_cfg45/prof_$w.log 2>&1) || { echo "rocprof $w failed"; tail -3 gpurun_out/r02_cfg45/prof_$w.log; }
  find gpurun_out/r02_cfg45/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -r head -4
done
date +%T
