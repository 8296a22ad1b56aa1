#!/bin/bash
# round 3: every row of DESIGN.md's table, default kernels
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
t --workload cfg3
t --workload cfg3 --pattern written
t --workload cfg3 --keep-off-diags
t --workload cfg3 --material miehe
t --workload cfg3 --mode accumulate
t --workload cfg3 --mode accumulate --pattern written
t --workload cfg3 --force-only
t --workload cfg2
t --workload cfg2 --n 4000
t --workload cfg2 --n 4000 --mode accumulate
t --workload tri3 --n 2000
t --workload tri3 --n 2000 --force-only
t --workload tri3g3 --n 2000
t --workload cfg3micro
t --workload cfg3micro --material bourdin
t --workload cfg3micro --material amor-split
t --workload cfg4 --n 128
t --workload cfg4 --n 128 --keep-off-diags
t --workload cfg4 --n 128 --force-only
t --workload cfg4 --n 128 --mode accumulate
t --workload cfg4 --n 96 --material miehe
t --workload cfg5 --n 64
t --workload cfg5g2 --n 64
t --workload cfg5 --n 64 --force-only
t --workload cfg5g2
t --workload quad8 --n 1000
t --workload quad8g2 --n 1000
t --workload quad8 --n 1000 --material miehe
t --workload tri6 --n 1000
t --workload tri6 --n 1000 --force-only
t --workload gd
t --workload gd --force-only
t --workload gd3 --n 128
t --workload gd3 --n 128 --force-only
t --workload cfg3 --material miehe --force-only
t --workload cfg3 --material miehe --mode accumulate
t --workload cfg4 --n 128 --material bourdin
t --workload cfg4 --n 128 --material bourdin --force-only
t --workload cfg4 --n 96 --material miehe --force-only
t --workload cfg4 --n 96 --material miehe --keep-off-diags
t --workload tri6 --n 1000 --material miehe
t --workload tri3g3 --n 2000 --material miehe
t --workload quad8 --n 1000 --force-only
t --workload cfg5
t --workload cfg4
t --workload cfg4 --force-only
t --workload cfg4 --mode accumulate
