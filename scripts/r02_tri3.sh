#!/bin/bash
# Triangle3 / Gauss1: the shape 45 of the reference's demo inputs use (MiehePhase 17, AmorPhase 10 of them)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; f=d.get('f_int_only',{}); print('Gel/s %.4f  kernel_ms %.3f  frac %.3f   f_int only: %.3f ms' % (d['value']/1e9, r['kernel_ms'], r['frac'], f.get('kernel_ms', float('nan'))))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
t --workload tri3 --n 2000
t --workload tri3 --n 2000 --material miehe
t --workload tri3 --n 2000 --keep-off-diags
t --workload tri3 --n 2000 --material miehe --keep-off-diags
t --workload tri3 --n 2000 --mode accumulate
t --workload tri3 --n 2000 --material miehe --mode accumulate
