#!/bin/bash
# round 3, batch 38: FA_ACCUMULATE passes of the Hex8 u-phi tile kernel: the row chunks of a lane's read-modify-write pulled into
# the L2 by LDS-DMA loads issued before the element loop (FA_TILE_ACC_TOUCH)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
V=$PWD/falcon_amd/variants/lib_touch.so
if [ -f $V ]; then
  export FA_LIB=$V
  timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/batch38_tests.txt 2>&1; rc=$?; tail -3 gpurun_out/batch38_tests.txt; [ $rc = 0 ] || exit 1
fi
for v in base touch; do
  if [ $v = base ]; then unset FA_LIB; else [ -f $V ] || continue; export FA_LIB=$V; fi
  echo "variant $v"
  t --workload cfg4 --n 128 --mode accumulate
  t --workload cfg4 --n 128 --mode accumulate --keep-off-diags
  t --workload cfg4 --n 128 --material bourdin --mode accumulate
  t --workload cfg4 --mode accumulate
done
