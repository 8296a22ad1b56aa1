#!/bin/bash
# after the AmorPhase split update joined the micromorphic branch: parity of the micromorphic cases + the general-tangent rows of the bench matrix
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "micro" --timeout 120 --timeout-method thread || exit 1
run() { timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 "$@" 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Gel/s %.4f  kernel_ms %.3f  frac %.3f' % (d['value']/1e9, r['kernel_ms'], r['frac']))" || { tail -5 gpurun_out/bench_err.txt; exit 1; }; }
t() { echo "== $*"; run "$@" || exit 1; }
t --workload quad8 --n 1000 --material miehe
t --workload cfg3 --material miehe
t --workload cfg3micro
t --workload cfg3micro --material bourdin
t --workload cfg3micro --material amor-split
t --workload cfg4 --n 96 --material miehe
