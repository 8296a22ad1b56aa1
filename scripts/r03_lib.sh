#!/bin/bash
# round 3 helpers: one bench line -> "kernel_ms frac" summary; every GPU step under its own time limit
B() {  # B <label> <bench args...>   (FA_LIB / env switches are taken from the environment)
  local label="$1"; shift
  local out
  out=$(timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 3 --no-f-int-only "$@" 2>gpurun_out/r03_err.txt | tail -1) || { echo "FAILED: $label"; tail -5 gpurun_out/r03_err.txt; return 1; }
  echo "$out" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-46s kernel_ms %8.3f  frac %.4f  Gel/s %.4f  setup %.1fs' % ('$label', r['kernel_ms'], r['frac'], d['value'] / 1e9, d['config']['setup_seconds']))"
}
