#!/bin/bash
# measurement only: Hex8 K pass with the Gauss-point records fetched instead of computed (results are NOT K).
# lib_exp.so was built from a scratch copy of falcon_amd/csrc in which phase 1 of assemble_tiles (Hex8, u-phi) was replaced by
#   for (i = tid; i < n_el * SLOTW; i += BLK) sRec[i] = P.K[element(i / SLOTW) * SLOTW + i % SLOTW] * 1e-30 + 1.0;
# (the patch is not kept in the tree: it computes nothing)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
for v in base exp; do
  if [ $v = base ]; then unset FA_LIB; else export FA_LIB=$PWD/falcon_amd/variants/lib_exp.so; fi
  echo "variant $v"
  t --workload cfg4 --n 128
  t --workload cfg4
done
