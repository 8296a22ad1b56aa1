#!/bin/bash
# round 3, batch 33: force-vector kernel on element tiles, 2-D shapes: workgroup size = elements per tile
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep -E "ELEMENT tiles" gpurun_out/bench_err.txt | head -1; }
t() { echo "== $*"; run "$@" || exit 1; }
export FA_VERBOSE=1
for b in 128 192 256 320; do
  echo "FA_FVEC_BLK=$b"; export FA_FVEC_BLK=$b
  t --workload cfg3 --force-only
  t --workload cfg3 --material miehe --force-only
  t --workload gd --force-only
  t --workload tri3 --n 2000 --force-only
  t --workload cfg2 --n 4000 --force-only
done
