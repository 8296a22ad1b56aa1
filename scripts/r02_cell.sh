#!/bin/bash
# cfg4 with box-cell tiles (FA_TILE_CELL) against the Morton cells
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep "tiles:" gpurun_out/bench_err.txt | tail -1; }
for c in 0 333 442 433 443 332; do
  echo "== cell $c"; FA_VERBOSE=1 FA_TILE_CELL=$c run --workload cfg4 --n 128 || exit 1
done
echo "== cell 333 keep"; FA_TILE_CELL=333 run --workload cfg4 --n 128 --keep-off-diags
echo "== cell 333 force-only"; FA_TILE_CELL=333 run --workload cfg4 --n 128 --force-only
