#!/bin/bash
# 3-D tile kernel: cfg4 at 100^3 and cfg5 at 64^3
timeout -k 10 200 python bench.py --workload cfg4 --n 100 --steps 5 --warmup 2 --cpu-elems 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 100^3', round(d['roofline']['kernel_ms'],3), 'ms frac', round(d['roofline']['frac'],4))" || exit 1
timeout -k 10 200 python bench.py --workload cfg5 --n 64 --steps 5 --warmup 2 --cpu-elems 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 64^3', round(d['roofline']['kernel_ms'],3), 'ms frac', round(d['roofline']['frac'],4))" || exit 1
