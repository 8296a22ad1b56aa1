#!/bin/bash
# Hex8 force-vector kernel: workgroup sizes and tile shapes (cfg4 at 128^3, GET_INT_VECTOR pass)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep "force-vector tiles:" gpurun_out/bench_err.txt | tail -1; }
echo "== tile kernel (FA_FVEC=0)"; FA_FVEC=0 run --workload cfg4 --n 128 --force-only || exit 1
for cfg in "128 0" "192 0" "256 0" "192 554" "192 455" "128 444" "256 655" "320 665"; do
  set -- $cfg
  echo "== blk $1 cell $2"; FA_VERBOSE=1 FA_FVEC_BLK=$1 FA_FVEC_CELL=$2 run --workload cfg4 --n 128 --force-only || exit 1
done
