#!/bin/bash
# round-2 starting point: the calls the reference makes on cfg3, default kernels and FA_TILES=1
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))"; }
ab() { echo "== $*"; echo -n "  default    "; run "$@" || exit 1; echo -n "  FA_TILES=1 "; FA_TILES=1 run "$@" || exit 1; }
ab --workload cfg3
ab --workload cfg3 --force-only
ab --workload cfg3 --mode accumulate
