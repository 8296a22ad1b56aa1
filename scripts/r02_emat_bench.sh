#!/bin/bash
# element-matrix kernel on cfg3 / cfg2: lanes per element and tile LDS budget
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
export FA_EMAT=1 FA_VERBOSE=1
for lpe in 2 1; do for lds in 81920 53000; do
  echo "== K lpe=$lpe lds=$lds"; FA_EMAT_LPE=$lpe FA_EMAT_LDS=$lds run --workload cfg3 || exit 1
done; done
grep element-matrix gpurun_out/bench_err.txt | tail -1
for lpe in 1 2; do for lds in 40000 81920; do
  echo "== f lpe=$lpe lds=$lds"; FA_EMAT_LPE_F=$lpe FA_EMAT_LDS=$lds run --workload cfg3 --force-only || exit 1
done; done
grep element-matrix gpurun_out/bench_err.txt | tail -1
echo "== accumulate"; run --workload cfg3 --mode accumulate
echo "== cfg2"; run --workload cfg2
echo "== cfg2 16M"; run --workload cfg2 --n 4000
