#!/bin/bash
# A/B of the tile kernel against the row / cooperative kernels: bench_tiles_ab.sh "<bench args>"...
one() { timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-elems 0 $1 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('[falcon'): print('   ', l.strip()[:150])
    if l.startswith('{'):
        d = json.loads(l); print('    kernel_ms %.3f  Gel/s %.4f  setup %.1fs' % (d['roofline']['kernel_ms'], d['value'] / 1e9, d['config']['setup_seconds']))
"; }
for a in "$@"; do
  echo "== $a"; echo "  tiles:"; FA_VERBOSE=1 FA_TILES=1 one "$a" || exit 1
  echo "  other:"; FA_TILES=0 one "$a" || exit 1
done
