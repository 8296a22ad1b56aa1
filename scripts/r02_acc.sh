#!/bin/bash
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f' % (d['value']/1e9, r['kernel_ms'], r['frac']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
echo "== cfg3 overwrite"; run --workload cfg3 || exit 1
echo "== cfg3 accumulate"; run --workload cfg3 --mode accumulate || exit 1
echo "== cfg3 accumulate written"; run --workload cfg3 --mode accumulate --pattern written || exit 1
echo "== cfg2 16M accumulate"; run --workload cfg2 --n 4000 --mode accumulate || exit 1
echo "== cfg3 force-only"; run --workload cfg3 --force-only || exit 1
echo "== cfg3 force-only accumulate"; run --workload cfg3 --force-only --mode accumulate || exit 1
