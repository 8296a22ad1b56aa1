#!/bin/bash
# round 3, batch 32: element tiles of the tile kernel's GET_INT_VECTOR passes, 2-D shapes: elements per tile
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep -E "ELEMENT tiles" gpurun_out/bench_err.txt | head -1; }
t() { echo "== $*"; run "$@" || exit 1; }
export FA_LIB=$PWD/falcon_amd/variants/lib_A.so
export FA_VERBOSE=1
for w in "tri6 --n 1000" "quad8 --n 1000" "quad8g2 --n 1000" "tri3g3 --n 2000" "quad8 --n 1000 --material miehe"; do
  echo "FA_ELEM_TILES=0"; FA_ELEM_TILES=0 t --workload $w --force-only
  for el in 64 128 256 512; do echo "FA_ELEM_TILE_EL=$el"; FA_ELEM_TILES=2 FA_ELEM_TILE_EL=$el t --workload $w --force-only; done
done
for el in 128 256; do echo "FA_ELEM_TILE_EL=$el"; FA_ELEM_TILES=2 FA_ELEM_TILE_EL=$el t --workload cfg5g2 --n 64 --force-only; done
