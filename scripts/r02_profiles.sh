#!/bin/bash
# usage: r02_profiles.sh TAG [bench args]   -> gpurun_out/r02_prof/TAG/{bench.json,stats,fetch,write}
# the round's evidence for one bench configuration: the JSON line, rocprofv3 kernel-trace statistics and the two HBM
# traffic counter passes (FETCH_SIZE, WRITE_SIZE), every profiler pass in a run of its own
export TMPDIR=/tmp; R=$PWD; tag=$1; shift; O=$R/gpurun_out/r02_prof/$tag; mkdir -p $O
B="python3 $R/bench.py --cpu-elems 0 --no-f-int-only $*"
timeout -k 10 400 $B --steps 10 --warmup 3 2> $O/bench.err | tail -1 > $O/bench.json || { echo "$tag: bench failed"; tail -3 $O/bench.err; exit 1; }
python3 -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print('$tag: kernel_ms %.3f frac %.4f value %.4g' % (r['kernel_ms'], r['frac'], d['value']))"
(cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 5 --warmup 2 > $O/stats.log 2>&1) || { echo "$tag: stats pass failed"; tail -3 $O/stats.log; exit 1; }
(cd /tmp && timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 3 --warmup 1 > $O/fetch.log 2>&1) || { echo "$tag: fetch pass failed"; tail -3 $O/fetch.log; exit 1; }
(cd /tmp && timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 3 --warmup 1 > $O/write.log 2>&1) || { echo "$tag: write pass failed"; tail -3 $O/write.log; exit 1; }
ls $O/stats/*/ | grep kernel_stats
