#!/bin/bash
# round 3, batch 34: FA_ACCUMULATE passes of the Hex8 tile kernel with K += as atomic adds without return (FA_TILE_ACC_ATOMIC=1)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
export FA_LIB=$PWD/falcon_amd/variants/lib_at.so
timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for v in base atomic; do
  if [ $v = base ]; then unset FA_LIB; else export FA_LIB=$PWD/falcon_amd/variants/lib_at.so; fi
  echo "variant $v"
  t --workload cfg4 --n 128 --mode accumulate
  t --workload cfg4 --n 128 --mode accumulate --keep-off-diags
  t --workload cfg4 --n 96 --material miehe --mode accumulate
  t --workload cfg4 --mode accumulate
done
