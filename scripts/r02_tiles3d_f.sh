#!/bin/bash
# 3-D tile kernel: phase 0 + 1 alone (force-only passes through the tile kernel)
for w in "cfg4 --n 100" "cfg5 --n 64"; do
  FA_TILES=1 timeout -k 10 200 python bench.py --workload $w --steps 5 --warmup 2 --cpu-elems 0 --force-only 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w force-only', round(d['roofline']['kernel_ms'],3), 'ms')" || exit 1
done
