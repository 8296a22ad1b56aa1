#!/bin/bash
# round-end evidence: GPU tests, smoke, default bench, kernel-trace stats (cfg3, cfg4), HBM traffic passes for cfg4
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
B3="python3 $R/bench.py --steps 10 --warmup 2 --cpu-elems 0"
B4="python3 $R/bench.py --steps 10 --warmup 2 --cpu-elems 0 --workload cfg4 --n 128"
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -- $B3 > $O/stats_cfg3.log 2>&1 && \
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- $B4 > $O/stats_cfg4.log 2>&1 && \
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfg4/fetch -- $B4 > $O/fetch_cfg4.log 2>&1 && \
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cfg4/write -- $B4 > $O/write_cfg4.log 2>&1
echo "profiles rc=$?"; ls $O/*/*/ | head -20
