#!/bin/bash
# usage: bench_variants.sh "<bench args>" variant...   -> one line per variant (kernel ms)
args="$1"; shift
for v in "$@"; do
  echo "== $v"
  FA_LIB=$PWD/variants/$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-elems 0 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('kernel_ms %.3f  Gel/s %.4f' % (d['roofline']['kernel_ms'], d['value'] / 1e9))
" || exit 1
done
