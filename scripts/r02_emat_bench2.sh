#!/bin/bash
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
export FA_EMAT=1 FA_VERBOSE=1
for lpe in 2 1; do
  echo "== K lpe=$lpe"; FA_EMAT_LPE=$lpe run --workload cfg3 || exit 1
done
echo "== K lpe=2 written"; FA_EMAT_LPE=2 run --workload cfg3 --pattern written
echo "== cfg2 16M lpe2"; FA_EMAT_LPE=2 run --workload cfg2 --n 4000
echo "== cfg2 16M lpe1"; FA_EMAT_LPE=1 run --workload cfg2 --n 4000
echo "== tri3 lpe1"; run --workload tri3 --n 2000
