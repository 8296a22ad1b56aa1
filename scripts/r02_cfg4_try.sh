#!/bin/bash
# cfg4 at 100^3: tile geometries
for big in 0 1; do
  echo "FA_TILE_BIG=$big"
  FA_VERBOSE=1 FA_TILE_BIG=$big timeout -k 10 200 python bench.py --workload cfg4 --n 100 --steps 5 --warmup 2 --cpu-elems 0 2>&1 | grep -E "tiles:|metric" | cut -c1-260 || exit 1
done
