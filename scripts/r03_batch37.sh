#!/bin/bash
# round 3, batch 37: FA_ACCUMULATE passes of the Hex8 u-phi tile kernel: the old values of a lane's own block as the initial
# values of its accumulators (requested before the element loop), the block then stored without a read
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
if [ -f falcon_amd/variants/lib_accpre.so ]; then
  export FA_LIB=$PWD/falcon_amd/variants/lib_accpre.so
  timeout -k 10 600 python -u -m pytest tests/test_gpu_tiles.py tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "tile or accumulate or phase_fracture" 2>&1 | tail -3 || exit 1
fi
for v in base accpre; do
  if [ $v = base ]; then unset FA_LIB; else [ -f falcon_amd/variants/lib_accpre.so ] || continue; export FA_LIB=$PWD/falcon_amd/variants/lib_accpre.so; fi
  echo "variant $v"
  t --workload cfg4 --n 128 --mode accumulate
  t --workload cfg4 --n 128 --material bourdin --mode accumulate
  t --workload cfg4 --mode accumulate
done
