#!/bin/bash
# round 3, batch 27: point-major Gauss-point lanes for Hex20 GET_INT_VECTOR passes
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f  GB/s %.0f  B/el %.0f setup %.1fs' % (d['value']/1e9, r['kernel_ms'], r['frac'], r['achieved'], r['bytes_per_element'], d['config']['setup_seconds']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
t --workload cfg5 --n 64 --force-only
t --workload cfg5g2 --n 64 --force-only
t --workload cfg5 --n 64 --material miehe --force-only
t --workload cfg5 --n 64 --material bourdin --force-only
