#!/bin/bash
# round 3, batch 29: Hex20 GET_INT_VECTOR passes -- lane order (A/B/D element-major, C/E point-major), node-loop unroll
# (A 1, B/C 4, D/E 20) and workgroup size
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
for v in C A B D E; do
  if [ $v = C ]; then unset FA_LIB; else export FA_LIB=$PWD/falcon_amd/variants/lib_$v.so; fi
  for fb in 512 1024; do
    echo "variant $v FA_TILE_FBLOCK=$fb"
    FA_TILE_FBLOCK=$fb t --workload cfg5 --n 64 --force-only
    FA_TILE_FBLOCK=$fb t --workload cfg5g2 --n 64 --force-only
  done
done
