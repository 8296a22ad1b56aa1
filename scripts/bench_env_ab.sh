#!/bin/bash
# usage: bench_env_ab.sh VAR "v1 v2 ..." "<bench args>"...: kernel ms for each value of an environment switch
var=$1; vals=$2; shift 2
for a in "$@"; do
  echo "== $a"
  for v in $vals; do
    echo -n "   $var=$v  "
    env $var=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-elems 0 $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('kernel_ms %.3f  Gel/s %.4f' % (d['roofline']['kernel_ms'], d['value'] / 1e9))
" || exit 1
  done
done
