#!/bin/bash
# round 3, batch 31: element tiles for the GET_INT_VECTOR passes of the 2-D shapes (FA_ELEM_TILES=2)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep -E "ELEMENT tiles" gpurun_out/bench_err.txt | head -1; }
t() { echo "== $*"; run "$@" || exit 1; }
export FA_LIB=$PWD/falcon_amd/variants/lib_A.so
export FA_VERBOSE=1
for et in 1 2; do
  echo "FA_ELEM_TILES=$et"; export FA_ELEM_TILES=$et
  t --workload cfg3 --force-only
  t --workload cfg3 --material miehe --force-only
  t --workload gd --force-only
  t --workload tri3 --n 2000 --force-only
  t --workload quad8 --n 1000 --force-only
  t --workload tri6 --n 1000 --force-only
  t --workload cfg2 --n 4000 --force-only
done
