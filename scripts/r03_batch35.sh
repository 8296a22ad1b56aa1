#!/bin/bash
# round 3, batch 35: force-vector kernel, Quad4 PhaseFracture / AmorPhase folded in (134 VGPRs): asked for four waves per SIMD
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 20 --warmup 3 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
for v in base cap4; do
  if [ $v = base ]; then unset FA_LIB; else export FA_LIB=$PWD/falcon_amd/variants/lib_c4.so; fi
  echo "variant $v"
  t --workload cfg3 --force-only
  FA_FVEC_BLK=128 t --workload cfg3 --force-only
  FA_FVEC_BLK=256 t --workload cfg3 --force-only
  FA_ELEM_TILES=0 t --workload cfg3 --force-only
done
