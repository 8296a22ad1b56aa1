#!/bin/bash
# round 3, batch 28: Hex20 GET_INT_VECTOR tiles -- tile statistics and sensitivity to the tile size
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; grep -E "force-only tiles" gpurun_out/bench_err.txt | head -1; }
t() { echo "== $*"; run "$@" || exit 1; }
export FA_VERBOSE=1
for fs in 64 128 256 512 1024; do echo "FA_TILE_FSET=$fs"; FA_TILE_FSET=$fs t --workload cfg5 --n 64 --force-only; done
FA_TILE_FSET=0 t --workload cfg5 --n 64 --force-only
for fs in 128 512; do echo "FA_TILE_FSET=$fs"; FA_TILE_FSET=$fs t --workload cfg5g2 --n 64 --force-only; done
