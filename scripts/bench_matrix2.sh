#!/bin/bash
# secondary workloads: Miehe material, off-diagonal blocks, hex20 (cfg5)
run() { echo "== $*"; timeout -k 10 400 python bench.py --cpu-elems 0 --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))"; }
run --workload cfg3 --material miehe
run --workload cfg3 --keep-off-diags
run --workload cfg3 --material miehe --force-only
run --workload cfg4 --n 96 --material miehe
run --workload cfg5 --n 48
run --workload cfg5g2 --n 48
