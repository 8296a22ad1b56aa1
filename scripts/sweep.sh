#!/bin/bash
# run bench.py against every experiment build in variants/ (one process each), print kernel ms
for so in variants/*.so; do
  r=$(FA_LIB=$PWD/$so timeout -k 10 200 python bench.py --cpu-elems 0 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))") || { echo "$so FAILED"; exit 1; }
  echo "$so $r"
done
