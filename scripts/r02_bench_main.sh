#!/bin/bash
# the calls the reference makes on cfg3 + cfg2, default kernels
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
echo "== cfg3 overwrite"; run --workload cfg3 || exit 1
echo "== cfg3 accumulate"; run --workload cfg3 --mode accumulate || exit 1
echo "== cfg3 force-only"; run --workload cfg3 --force-only || exit 1
echo "== cfg3 written"; run --workload cfg3 --pattern written || exit 1
echo "== cfg3 keep"; run --workload cfg3 --keep-off-diags || exit 1
echo "== cfg3 miehe"; run --workload cfg3 --material miehe || exit 1
echo "== cfg2"; run --workload cfg2 || exit 1
echo "== cfg2 16M"; run --workload cfg2 --n 4000 || exit 1
echo "== tri3 8M"; run --workload tri3 --n 2000 || exit 1
