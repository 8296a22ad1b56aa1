#!/bin/bash
# cfg3 headline only (overwrite, GET_MATRIX0)
timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('kernel_ms %.3f frac %.3f' % (r['kernel_ms'], r['frac']))"
