#!/bin/bash
# 3-D tile kernel: pair ownership across tiles x lane groups
for sym in 0 1; do for sp in 0 1 2; do
  echo "FA_TILE_SYM=$sym FA_TILE_SPLIT=$sp"
  FA_TILE_SYM=$sym FA_TILE_SPLIT=$sp ./scripts/r02_tiles3d.sh || exit 1
done; done
