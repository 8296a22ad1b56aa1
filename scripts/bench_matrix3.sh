#!/bin/bash
# 3-D and higher-order workloads on the tile kernel (default policy), with the kernel it replaced beside it (FA_TILES=0)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))"; }
ab() { echo "== $*"; echo -n "  default   "; run "$@" || exit 1; echo -n "  FA_TILES=0 "; FA_TILES=0 run "$@" || exit 1; }
ab --workload cfg4 --n 128
ab --workload cfg4 --n 128 --keep-off-diags
ab --workload cfg4 --n 128 --force-only
ab --workload cfg4 --n 128 --mode accumulate
ab --workload cfg4 --n 96 --material miehe
ab --workload cfg5 --n 48
ab --workload cfg5g2 --n 48
ab --workload cfg5 --n 48 --force-only
ab --workload quad8 --n 1000
ab --workload quad8g2 --n 1000
ab --workload quad8 --n 1000 --material miehe
ab --workload tri6 --n 1000
ab --workload tri6 --n 1000 --force-only
