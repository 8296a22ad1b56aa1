#!/bin/bash
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f' % (d['value']/1e9, r['kernel_ms'], r['frac']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
for blk in 128 192 256; do
  echo "== default lib blk=$blk"; FA_FVEC_BLK=$blk run --workload cfg3 --force-only || exit 1
  for v in fvec_w4 fvec_w5; do echo "== $v blk=$blk"; FA_LIB=$PWD/variants/$v.so FA_FVEC_BLK=$blk run --workload cfg3 --force-only || exit 1; done
done
echo "== persist blk=192"; FA_PERSIST=1 run --workload cfg3 --force-only
