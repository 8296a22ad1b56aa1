#!/bin/bash
run() { timeout -k 10 300 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
export FA_VERBOSE=1
for blk in 192 256; do for per in 1 0; do
  echo "== fvec blk=$blk persist=$per"; FA_FVEC_BLK=$blk FA_PERSIST=$per run --workload cfg3 --force-only || exit 1
done; done
grep force-vector gpurun_out/bench_err.txt | tail -1
echo "== fvec accumulate"; run --workload cfg3 --force-only --mode accumulate
echo "== fvec miehe"; run --workload cfg3 --force-only --material miehe
echo "== fvec cfg2"; run --workload cfg2 --n 4000 --force-only
echo "== fvec tri3"; run --workload tri3 --n 2000 --force-only
echo "== FA_FVEC=0 tri3"; FA_FVEC=0 run --workload tri3 --n 2000 --force-only
