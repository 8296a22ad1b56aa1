#!/bin/bash
# bench.py over the modes reported in DESIGN.md (one process each)
run() { echo "== $*"; timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.3f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element']))"; }
run --workload cfg3
run --workload cfg3 --pattern written
run --workload cfg3 --mode accumulate
run --workload cfg3 --force-only
run --workload cfg2
run --workload cfg2 --n 4000
run --workload cfg4 --n 128
