#!/bin/bash
# MicroPhaseFracture with the model folded in at compile time on the lane-pair kernel: parity, then the bench rows
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "micro" --timeout 120 --timeout-method thread || exit 1
for m in miehe bourdin amor-split; do
timeout -k 10 400 python bench.py --cpu-elems 0 --steps 10 --warmup 2 --workload cfg3micro --material $m 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg3micro $m: kernel_ms %.3f frac %.3f' % (r['kernel_ms'], r['frac']))" || exit 1
done
