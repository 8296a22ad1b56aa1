#!/bin/bash
# the tile-kernel rows of the bench matrix (regression check after changes to the Gauss-point core)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
t --workload tri3g3 --n 2000
t --workload cfg4 --n 128
t --workload cfg4 --n 128 --keep-off-diags
t --workload cfg4 --n 128 --force-only
t --workload cfg4 --n 128 --mode accumulate
t --workload quad8 --n 1000
t --workload tri6 --n 1000
t --workload tri6 --n 1000 --force-only
t --workload cfg3micro --material bourdin
