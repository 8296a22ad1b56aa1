#!/bin/bash
# round 3, batch 39: Hex20 grouped pair lanes: the neighbour words read after the element loop instead of in front of it
# (four registers less across the loop: scratch of the 1024-thread Gauss3^3 kernel 20 -> 12 bytes)
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
V=$PWD/falcon_amd/variants/lib_bw.so
for v in base bw; do
  if [ $v = base ]; then unset FA_LIB; else [ -f $V ] || continue; export FA_LIB=$V; fi
  echo "variant $v"
  t --workload cfg5 --n 64
  t --workload cfg5g2 --n 64
  t --workload cfg5
done
