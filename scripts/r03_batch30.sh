#!/bin/bash
# round 3, batch 30: element tiles for the GET_INT_VECTOR passes of the 3-D shapes
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep -E "ELEMENT tiles" gpurun_out/bench_err.txt | head -1; }
t() { echo "== $*"; run "$@" || exit 1; }
export FA_LIB=$PWD/falcon_amd/variants/lib_A.so
timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
export FA_VERBOSE=1
for et in 0 1; do
  echo "FA_ELEM_TILES=$et"; export FA_ELEM_TILES=$et
  t --workload cfg4 --n 128 --force-only
  t --workload cfg4 --n 96 --material miehe --force-only
  t --workload cfg4 --n 128 --material bourdin --force-only
  t --workload gd3 --n 128 --force-only
  t --workload cfg5 --n 64 --force-only
  t --workload cfg5g2 --n 64 --force-only
done
export FA_ELEM_TILES=1
for el in 32 96 128; do echo "FA_ELEM_TILE_EL=$el"; FA_ELEM_TILE_EL=$el t --workload cfg5 --n 64 --force-only; done
for b in 192 256; do echo "FA_FVEC_BLK=$b";  FA_FVEC_BLK=$b t --workload cfg4 --n 128 --force-only; done
t --workload cfg4 --force-only
